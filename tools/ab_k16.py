"""encode time of the trainer's first-phase shape (16 x 16 codebooks, dim 512) at a trainer batch and at 65,536 vectors;
MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=... python tools/ab_k16.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantization_amd import Quantizer, synthetic as gen  # noqa: E402

D, K, N = 512, 16, 16
sd = gen.synthetic_state(5, D, K, N)
q = Quantizer(D, K, N)
st = q.state_dict()
for k, v in sd.items():
    st[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(st)
q = q.cuda()
out = []
for B, reps in ((4096, 300), (65536, 30)):
    x = torch.randn(B, D, device="cuda")
    with torch.no_grad():
        for _ in range(5):
            q.encode(x, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            q.encode(x, 2)
        e1.record()
        torch.cuda.synchronize()
    out.append((B, round(e0.elapsed_time(e1) / reps, 4)))
print(os.path.basename(os.environ.get("MCQ_LIB_PATH", "libmcq_hip.so")), out)
