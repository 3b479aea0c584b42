"""decode of one shape under rocprofv3: 60 mcq_decode launches of `batch` vectors (usage: exp_decode_shape.py dim ncb batch)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import synthetic as gen  # noqa: E402
from bench import load_quantizer  # noqa: E402

D, N, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
q = load_quantizer(gen.synthetic_state(103, D, 256, N), D, 256, N, dev)
codes = torch.randint(0, 256, (B, N), dtype=torch.uint8, device=dev)
with torch.no_grad():
    for _ in range(60):
        y = q.decode(codes)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(50):
        y = q.decode(codes)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"decode dim {D}, {N} x 256, {B} vectors: {ms * 1e3:.1f} us per call, {B * (N + 4 * D) / ms / 1e6:.1f} GB/s")
