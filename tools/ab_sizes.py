"""encode time of the headline shape at several batch sizes (one build per call): python tools/ab_sizes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantization_amd import Quantizer, synthetic as gen  # noqa: E402

D, K, N = 512, 256, 8
sd = gen.synthetic_state(103, D, K, N)
q = Quantizer(D, K, N)
st = q.state_dict()
for k, v in sd.items():
    st[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(st)
q = q.cuda()
out = []
for B, reps in ((64, 200), (4096, 100), (65536, 20)):
    x = torch.randn(B, D, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            q.encode(x, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            q.encode(x, 5)
        e1.record()
        torch.cuda.synchronize()
    out.append((B, round(e0.elapsed_time(e1) / reps, 4)))
print(os.path.basename(ROOT), out)
