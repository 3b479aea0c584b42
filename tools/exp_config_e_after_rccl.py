"""Does a one-rank RCCL group that was created (and destroyed) earlier in the process slow the host-bound trainer step down?
config E (20,001 steps) timed twice before and twice after."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from quantization_amd import QuantizerTrainer
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
def run(tag):
    random.seed(0); torch.manual_seed(0)
    tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=dev)
    gq = torch.Generator(device=dev); gq.manual_seed(1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while not tr.done():
        tr.step(torch.randn(4096, 512, device=dev, generator=gq)); n += 1
    torch.cuda.synchronize()
    print(f"{tag}: {time.perf_counter() - t0:.3f} s for {n} steps", flush=True)
run("cold"); run("warm")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
run("rccl group alive")
dist.destroy_process_group()
run("rccl group destroyed")
