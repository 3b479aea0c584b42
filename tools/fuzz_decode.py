"""Random-shape differential run of mcq_decode on LARGE batches (the block-staged LDS-resident kernel and its fall-backs) against
torch's gather + in-order sum on the device, every row, bit for bit:  python tools/fuzz_decode.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import Quantizer, synthetic as gen  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    K = int(rs.choice([32, 64, 128, 256]))
    N = int(rs.choice([2, 4, 8, 16]))
    D = int(rs.choice([4 * rs.randint(1, 20), 4 * rs.randint(20, 130), 4 * rs.randint(130, 300), rs.randint(1, 600)]))
    B = int(rs.choice([rs.randint(16384, 20000), rs.randint(20000, 70000), rs.randint(70000, 300000)]))
    B = min(B, int(1.5e9 // (4 * D)))
    sd = gen.synthetic_state(3000 + c, D, K, N)
    q = Quantizer(D, K, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    with torch.no_grad():
        codes = torch.randint(0, K, (B + 1, N), device="cuda", dtype=torch.uint8)
        C = q.get_centers()
        def ref(cd):
            acc = C[0][cd[:, 0].long()]
            for n in range(1, N):
                acc = acc + C[n][cd[:, n].long()]
            return acc
        ok = True
        for off in (0, 1):          # aligned codes; codes at an odd byte offset (another kernel)
            cd = codes[off:off + B]
            y = q.decode(cd)
            ok &= bool(torch.equal(y, ref(cd)))
        y64 = q.decode(codes[:B].long())
        ok &= bool(torch.equal(y64, ref(codes[:B])))
    print(f"case {c}: D={D} K={K} N={N} B={B} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del q, codes, y, y64, C
    torch.cuda.empty_cache()
print("cases", cases, "mismatching", bad)
sys.exit(1 if bad else 0)
