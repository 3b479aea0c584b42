"""Repeat the headline encode (and two other shapes) many times and compare every result with the first: a race in the hand-
counted waits of the product kernel or in an LDS hand-off would show as a flipped code.  python tools/soak.py [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantization_amd import Quantizer, synthetic as gen  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (D, K, N, B) in [(512, 256, 8, 65536), (1024, 256, 16, 16384), (200, 64, 4, 30000), (512, 16, 16, 20000)]:
    sd = gen.synthetic_state(3, D, K, N)
    q = Quantizer(D, K, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    x = torch.randn(B, D, device="cuda")
    with torch.no_grad():
        ref = q.encode(x, 5)
        dref = q.decode(ref)
        n = reps if D == 512 and K == 256 else max(reps // 6, 20)
        diff = 0
        for i in range(n):
            if i % 7 == 3:
                q.invalidate_cache()          # rebuild the derived state (Gram matrix through the same kernel)
            c = q.encode(x, 5)
            diff += int((c != ref).any(dim=1).sum())
            if i % 10 == 0:
                diff += int((q.decode(c) != dref).any(dim=1).sum())
        torch.cuda.synchronize()
    print(f"D={D} K={K} N={N} B={B}: {n} repeats, differing vectors {diff}", flush=True)
    bad += diff
print("total differing", bad)
sys.exit(1 if bad else 0)
