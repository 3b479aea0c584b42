"""cProfile of the host side of QuantizerTrainer.step (phase 1 at config E's shape): where the 0.25 ms of enqueue time per step go"""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import QuantizerTrainer
dev = torch.device("cuda:0")
phase2 = len(sys.argv) > 1 and sys.argv[1] == "2"
tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=dev, phase_one_iters=(5 if phase2 else 100000), phase_two_iters=100000)
x = torch.randn(4096, 512, device=dev)
for _ in range(30):
    tr.step(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    tr.step(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
