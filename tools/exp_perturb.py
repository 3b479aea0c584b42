"""How many codes move under an ulp-scale (1e-7 relative) perturbation of the parameters: the sensitivity behind the chaotic drift of
two trainer runs whose gradient sums differ in their last bits (tests/test_gpu_trainer_config_e.py, two ranks vs one process)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import gen
from quantization_amd import Quantizer
dev = torch.device("cuda:0")
torch.manual_seed(5)
for (D, K, N) in ((512, 16, 16), (512, 256, 8)):
    q = Quantizer(D, K, N).to(dev)
    x = torch.from_numpy(gen.make_x(3, 4096, D)).to(dev)
    for it in (1, 2):
        with torch.no_grad():
            base = q.encode(x, it, as_bytes=False).cpu().numpy()
            tot = 0
            for rep in range(5):
                q2 = Quantizer(D, K, N).to(dev)
                q2.load_state_dict(q.state_dict())
                g = torch.Generator(device=dev); g.manual_seed(rep)
                q2.centers.mul_(1.0 + 1e-7 * torch.randn(q2.centers.shape, device=dev, generator=g))
                q2.to_logits.weight.mul_(1.0 + 1e-7 * torch.randn(q2.to_logits.weight.shape, device=dev, generator=g))
                got = q2.encode(x, it, as_bytes=False).cpu().numpy()
                tot += int((got != base).any(axis=1).sum())
        print(D, K, N, "iters", it, "rows moved by a 1e-7 perturbation, mean of 5:", tot / 5)
