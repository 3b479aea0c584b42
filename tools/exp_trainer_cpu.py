"""Is the trainer step bound by host-side launch work?  Enqueue time vs completion time of 100 steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import QuantizerTrainer
for phase2 in (False, True):
    tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=torch.device("cuda"),
                          phase_one_iters=(5 if phase2 else 100000), phase_two_iters=100000)
    x = torch.randn(4096, 512, device="cuda")
    for _ in range(12):
        tr.step(x)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("phase", 2 if phase2 else 1, "enqueue ms/step %.3f   done ms/step %.3f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
