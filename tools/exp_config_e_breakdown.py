"""Where config E's wall time goes beside its 20,001 free-running steps: the reference's progress log (every 200 iterations:
six compute_loss calls with a .item() each, quantization.py:656-671) and the frame generation.
Runs: as shipped; with the log's compute_loss replaced by ready device tensors (what remains is the pipeline drain of .item());
with CPU tensors (no drain either); without generating frames (one fixed batch)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_amd import QuantizerTrainer

dev = torch.device("cuda:0")
D, N = 512, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000

def run(mode, fixed_batch=False):
    random.seed(0); torch.manual_seed(0)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=N, device=dev, phase_one_iters=steps, phase_two_iters=steps)
    gq = torch.Generator(device=dev); gq.manual_seed(1)
    xfix = torch.randn(4096, D, device=dev, generator=gq)
    def patch():
        q = tr.quantizer
        if mode == "dev":
            z = torch.zeros((), device=dev)
            q.compute_loss = lambda x, j=0: (z, z, z, z)
        elif mode == "cpu":
            z = torch.zeros(())
            q.compute_loss = lambda x, j=0: (z, z, z, z)
    patch()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; q0 = tr.quantizer
    while not tr.done():
        if tr.quantizer is not q0:      # the second phase has its own module
            q0 = tr.quantizer; patch()
        tr.step(xfix if fixed_batch else torch.randn(4096, D, device=dev, generator=gq)); n += 1
    torch.cuda.synchronize()
    return time.perf_counter() - t0, n

for mode, fixed in (("as_shipped", False), ("dev", False), ("cpu", False), ("as_shipped", True)):
    t, n = run(mode, fixed)
    print(f"log={mode} fixed_batch={fixed}: {t:.3f} s for {n} steps = {t / n * 1e3:.4f} ms per step", flush=True)
