"""tests/golden/loss_k512.npz: Quantizer.compute_loss of the REFERENCE (quantization.py:184-242) and its gradients for a
quantizer of 4 codebooks of 512 entries -- a shape QuantizerTrainer never produces and the fused loss kernels do not cover, so
the product takes its torch-op formulation there.  Runs only in the build container (imports /root/reference)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")
import quantization as refq  # noqa: E402

import gen  # noqa: E402

D, K, N, B = 32, 512, 4, 384
sd = gen.synthetic_state(61, D, K, N)
q = refq.Quantizer(dim=D, codebook_size=K, num_codebooks=N)
st = q.state_dict()
for k, v in sd.items():
    st[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(st)
x = torch.from_numpy(gen.make_x(62, B, D))
out = {"D": D, "K": K, "N": N, "B": B, "state_seed": 61, "x_seed": 62}
for iters in (0, 2):
    q.zero_grad()
    losses = q.compute_loss(x, iters)
    tot = losses[0] + 0.3 * losses[1] + 0.2 * losses[2] + 0.1 * losses[3]      # (index entropy carries no gradient)
    tot.backward()
    out[f"losses_it{iters}"] = np.array([float(v) for v in losses], np.float64)
    for name, p in q.named_parameters():
        out[f"grad_it{iters}.{name}"] = p.grad.detach().numpy().copy()
    print(iters, out[f"losses_it{iters}"])
np.savez_compressed(os.path.join(HERE, "loss_k512.npz"), **out)
