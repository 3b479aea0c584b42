"""Trainer trajectory fixture: the reference QuantizerTrainer on CPU,
dim=64, bytes_per_frame=4, batch 256, phase_one_iters = phase_two_iters = 6, seeded.
Runs only where /root/reference is importable.  Stores per-step losses and learning
rate, the quantizer shape per step, and the final parameters."""
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

D, BYTES, BATCH, P1, P2, SEED = 64, 4, 256, 6, 6, 5
NAME, DATA_SEED, KEEP_FINAL = "trainer_d64_b4", 9000, True
if len(sys.argv) > 1 and sys.argv[1] == "config_e":
    # BASELINE.json config E at its own shape: dim 512, 8 bytes per frame, batches of 4,096 frames; 12 + 12 iterations.
    # The final state (8.4 MB) is not stored: every 37th row of the centers and of the classifier is.
    D, BYTES, BATCH, P1, P2, SEED = 512, 8, 4096, 12, 12, 6
    NAME, DATA_SEED, KEEP_FINAL = "trainer_config_e_d512_b8", 9500, False


def main():
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = refq.QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1,
                               phase_two_iters=P2)
    init = {k: v.detach().numpy().copy() for k, v in tr.quantizer.state_dict().items()}
    losses, lrs, shapes, two_iter = [], [], [], []
    orig = refq.Quantizer.compute_loss
    rec = {}

    def spy(self, x, refine_indexes_iters=0):
        out = orig(self, x, refine_indexes_iters)
        rec.setdefault("calls", []).append((refine_indexes_iters, [float(v) for v in out]))
        return out

    refq.Quantizer.compute_loss = spy
    it = 0
    while not tr.done():
        rec["calls"] = []
        shapes.append((tr.quantizer.codebook_size, tr.quantizer.num_codebooks))
        lrs.append(tr.optim.param_groups[0]["lr"])
        tr.step(torch.from_numpy(gen.make_x(DATA_SEED + it, BATCH, D)))
        iters, vals = rec["calls"][0]          # the training call is the first compute_loss of the step
        two_iter.append(iters)
        losses.append(vals)
        it += 1
        print("step", it, vals, flush=True)
    refq.Quantizer.compute_loss = orig
    final = {k: v.detach().numpy().copy() for k, v in tr.get_quantizer().state_dict().items()}
    out = {"D": D, "bytes": BYTES, "batch": BATCH, "P1": P1, "P2": P2, "seed": SEED, "steps": it,
           "losses": np.array(losses, np.float64), "lr": np.array(lrs, np.float64),
           "shapes": np.array(shapes, np.int64), "refine_iters": np.array(two_iter, np.int64)}
    for k, v in init.items():
        out["init." + k] = v
    out["data_seed"] = DATA_SEED
    for k, v in final.items():
        if KEEP_FINAL or v.ndim == 0:
            out["final." + k] = v
        elif k != "id_buf":
            out["final_rows37." + k] = v.reshape(-1, v.shape[-1])[::37].copy() if v.ndim >= 2 else v[::37].copy()
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), **out)
    print("steps", it, "shapes", shapes, "lr", lrs)
    print("losses[0]", losses[0], "losses[-1]", losses[-1])


if __name__ == "__main__":
    main()
