"""How many vectors are already at a fixed point of _refine_indexes after t passes?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from golden import fixtures
from quantization_amd import Quantizer

def load(st, D, K, N):
    q = Quantizer(D, K, N); sd = q.state_dict()
    for k, v in st.items(): sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); return q.cuda()

def report(name, q, x):
    with torch.no_grad():
        prev = q.encode(x, 0, as_bytes=False)
        out = []
        for t in range(1, 8):
            cur = q.encode(x, t, as_bytes=False)
            out.append(float((cur != prev).any(dim=1).float().mean()))
            prev = cur
    print(name, "fraction of vectors changed by pass t=1..7:", [round(v, 4) for v in out])

D, N, K = 512, 8, 256
report("bench synthetic (random codebooks)", load(gen.synthetic_state(103, D, K, N), D, K, N), torch.randn(65536, D, device="cuda"))
for name in ("trained_d64_b8_p2", "trained_d64_b4_p2", "trained_d64_b8_p1"):
    fx = fixtures.load(name)
    report(name, load(fx["state"], fx["D"], fx["K"], fx["N"]), torch.from_numpy(fx["x"]).cuda())
