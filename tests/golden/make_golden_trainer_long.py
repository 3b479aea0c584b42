"""Floor for the full-length config-E test (tests/test_gpu_trainer_long.py): the REFERENCE trainer at config E's shape
(dim 512, 8 bytes per frame) on Gaussian frames for a SHORT schedule (400 + 400 iterations of 600 frames, CPU), its losses
every 50th step and the relative reconstruction error of the quantizer it ends with on 30 held-out batches -- the measure of the
reference's own integration tests (test_quantization.py:41-46).  A run of the full 10,000 + 10,000 iterations must end below it.

    python tests/golden/make_golden_trainer_long.py        (about ten minutes on 8 cores)
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")
import quantization as refq  # noqa: E402

DIM, BYTES, BATCH, P1, P2, SEED = 512, 8, 600, 400, 400, 5


def main():
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = refq.QuantizerTrainer(dim=DIM, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1, phase_two_iters=P2)
    g = torch.Generator()
    g.manual_seed(SEED + 1)
    losses = []
    while not tr.done():
        x = torch.randn(BATCH, DIM, generator=g)
        if tr.cur_iter % 50 == 0:      # every 50th step: the four losses of compute_loss on the batch the step sees (:696-706)
            with torch.no_grad():
                rl, ll, le, ie = tr.quantizer.compute_loss(x, 1)
            losses.append([tr.cur_iter, float(rl), float(ll), float(le), float(ie)])
            print(f"   step {tr.cur_iter}: reconstruction loss {losses[-1][1]:.4f}", flush=True)
        tr.step(x)
    q = tr.get_quantizer()
    mean = q.get_data_mean()
    err = 0.0
    with torch.no_grad():
        for _ in range(30):
            x = torch.randn(BATCH, DIM, generator=g)
            err += float(((x - q.decode(q.encode(x))) ** 2).sum() / ((x - mean) ** 2).sum()) / 30
    print(f"reference, {P1} + {P2} iterations: relative reconstruction error {err:.5f} (Shannon bound {2 ** -(2 * BYTES * 8 / DIM):.5f})")
    np.savez_compressed(os.path.join(HERE, "trainer_long_d512_b8.npz"), dim=DIM, bytes=BYTES, batch=BATCH, p1=P1, p2=P2, seed=SEED,
                        losses=np.array(losses, np.float32), avg_rel_err=err, steps=len(losses))


if __name__ == "__main__":
    main()
