"""Runs QuantizerTrainer steps (dim 512, 8 bytes, batch 4096) in one phase for a kernel-trace profile."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import QuantizerTrainer
phase2 = len(sys.argv) > 1 and sys.argv[1] == "2"
tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=torch.device("cuda"),
                      phase_one_iters=(5 if phase2 else 100000), phase_two_iters=100000)
torch.manual_seed(0)
x = torch.randn(4096, 512, device="cuda")
for _ in range(10):
    tr.step(x)
torch.cuda.synchronize()
t = time.perf_counter()
n = 100
for _ in range(n):
    tr.step(x)
torch.cuda.synchronize()
print("phase", 2 if phase2 else 1, "ms/step", (time.perf_counter() - t) / n * 1e3)
