"""Is a QuantizerTrainer.step bound by the GPU or by the host that enqueues it?  Per phase at config E's shape: ms per step of the enqueue
loop alone (timed to the end of the Python loop, the device still busy) and of the whole (synchronised at the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import QuantizerTrainer
dev = torch.device("cuda:0")
for phase2 in (False, True):
    tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=dev, phase_one_iters=(5 if phase2 else 100000), phase_two_iters=100000)
    torch.manual_seed(0)
    x = torch.randn(4096, 512, device=dev)
    for _ in range(30):
        tr.step(x)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("phase", 2 if phase2 else 1, "enqueue %.3f ms/step, with the device drained %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
