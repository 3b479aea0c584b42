"""k_tf_pass16 against the separate launches at LARGE batches of 16-entry codebooks (the kernel was sized on the trainer's 4,096 frames)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import Quantizer, synthetic as gen
for N in (8, 16):
    sd = gen.synthetic_state(7, 512, 16, N)
    q = Quantizer(512, 16, N); st = q.state_dict()
    for k, v in sd.items(): st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st); q = q.cuda()
    for B in (4096, 16384, 65536, 262144):
        xg = torch.randn(B, 512, device="cuda")
        with torch.no_grad():
            for it in (1, 5):
                for _ in range(5): q.encode(xg, it)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30): q.encode(xg, it)
                torch.cuda.synchronize()
                print(f"{os.environ.get('AB_TAG','')} N={N} B={B} passes={it}: {(time.perf_counter()-t0)/30*1e6:.1f} us", flush=True)
