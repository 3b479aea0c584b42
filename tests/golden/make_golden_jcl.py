"""Generates tests/golden/jcl_*.npz: JointCodebookLoss of the reference (quantization/prediction.py:9-172) on seeded
inputs -- its initial state, the loss, and the gradients w.r.t. every parameter and the predictor.  Runs only in
the build container (imports the reference); the fixtures are data.

    python tests/golden/make_golden_jcl.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")
import quantization as refq  # noqa: E402


def gen(name, seed, pc, ncb, hidden, K, B, reduction, lead=None):
    torch.manual_seed(seed)
    m = refq.JointCodebookLoss(predictor_channels=pc, num_codebooks=ncb, hidden_channels=hidden, codebook_size=K,
                               reduction=reduction, checkpoint=False)
    with torch.no_grad():
        m.linear2_bias.normal_(std=0.1)      # zeros at init: make the bias path visible
    shape = (B,) if lead is None else lead
    pred = torch.randn(*shape, pc, requires_grad=True)
    idx = torch.randint(0, K, (*shape, ncb))
    flat = idx.reshape(-1, ncb)
    flat[::7] = -100                          # padding frames: all codebooks negative (:150-154)
    loss = m(pred, idx)
    loss.backward()
    out = dict(pc=pc, ncb=ncb, hidden=hidden, K=K, reduction=reduction, predictor=pred.detach().numpy(),
               indexes=idx.numpy(), loss=float(loss), grad_predictor=pred.grad.numpy())
    for k, v in m.state_dict().items():
        out["state." + k] = v.numpy()
    for k, p in m.named_parameters():
        out["grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"jcl_{name}.npz"), **out)
    print(name, "loss", float(loss))


if __name__ == "__main__":
    gen("small_k16", 3, 48, 4, 32, 16, 70, "sum")
    gen("k256_n4", 4, 32, 4, 48, 256, 96, "sum", lead=(4, 24))
    gen("mean_k64", 5, 40, 2, 64, 64, 50, "mean")
