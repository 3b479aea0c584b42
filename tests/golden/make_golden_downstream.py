"""The reference's downstream scenario (/root/reference/quantization/test_train_hdf5.py:79-134): a trained Quantizer encodes
frames, a JointCodebookLoss learns to predict the codes from the frames.  Runs the REFERENCE on CPU (quantizer trained by the
reference's trainer, predictor trained by torch.optim.Adam as in that script) on seeded frames and stores: the trained
quantizer, the predictor's initial state, the per-step loss of the predictor.  Runs only in the build container.

    python tests/golden/make_golden_downstream.py
"""
import os
import random
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

D, BYTES, QB, P1, P2, SEED = 64, 4, 256, 150, 150, 21      # the quantizer
B, STEPS, DATA_SEED = 512, 300, 30000                      # the predictor (test_train_hdf5.py:90, :112-131)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = refq.QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1, phase_two_iters=P2)
    it = 0
    while not tr.done():
        tr.step(torch.from_numpy(gen.make_x(20000 + it, QB, D)))
        it += 1
    q = tr.get_quantizer()
    torch.manual_seed(SEED + 1)
    predictor = refq.JointCodebookLoss(predictor_channels=D, num_codebooks=BYTES)
    init = {k: v.detach().numpy().copy() for k, v in predictor.state_dict().items()}
    optim = torch.optim.Adam(predictor.parameters(), lr=0.001, betas=(0.9, 0.98), eps=1e-9, weight_decay=1.0e-06)
    scheduler = torch.optim.lr_scheduler.StepLR(optim, step_size=2000, gamma=0.5)
    losses = []
    for s in range(STEPS):
        x = torch.from_numpy(gen.make_x(DATA_SEED + s, B, D))
        with torch.no_grad():
            encoding = q.encode(x)
        loss = predictor(x, encoding) / x.shape[0]
        losses.append(float(loss))
        loss.backward()
        optim.step()
        optim.zero_grad()
        scheduler.step()
        if s % 50 == 0:
            print(s, float(loss), flush=True)
    out = {"D": D, "bytes": BYTES, "B": B, "steps": STEPS, "data_seed": DATA_SEED, "losses": np.array(losses, np.float64)}
    for k, v in q.state_dict().items():
        out["quantizer." + k] = v.detach().numpy()
    for k, v in init.items():
        out["predictor_init." + k] = v
    np.savez_compressed(os.path.join(HERE, "downstream_d64_b4.npz"), **out)
    print("first", losses[0], "last 20 mean", float(np.mean(losses[-20:])))


if __name__ == "__main__":
    main()
