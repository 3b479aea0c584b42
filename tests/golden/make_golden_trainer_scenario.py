"""Trainer trajectory at the size of the reference's own integration scenario
(/root/reference/quantization/test_quantization.py:11-48: dim 256, 4 bytes per frame, 500 + 500 iterations, batches of
600 frames of "small random MLP of Gaussian noise + 0.05 * noise").  Runs the REFERENCE QuantizerTrainer on CPU, only
where /root/reference is importable; stores per-step losses, learning rate, refine-iteration draws and the final mean
relative reconstruction error on held-out batches.  The input generator (scenario_batch below: numpy float64 MLP with
seeded weights, rounded to fp32) is shared with the test, so no data is stored."""
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

from trainer_scenario import DIM, BYTES, BATCH, P1, P2, SEED, scenario_batch  # noqa: E402


def main():
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = refq.QuantizerTrainer(dim=DIM, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1,
                               phase_two_iters=P2)
    init = {k: v.detach().numpy().copy() for k, v in tr.quantizer.state_dict().items()}
    losses, lrs, two_iter = [], [], []
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
        lrs.append(tr.optim.param_groups[0]["lr"])
        tr.step(torch.from_numpy(scenario_batch(it)))
        iters, vals = rec["calls"][0]
        two_iter.append(iters)
        losses.append(vals)
        it += 1
        if it % 100 == 0:
            print(it, vals, flush=True)
    refq.Quantizer.compute_loss = orig
    q = tr.get_quantizer()
    mean = q.get_data_mean()
    rel = 0.0
    with torch.no_grad():
        for i in range(30):                                   # test_quantization.py:41-46
            x = torch.from_numpy(scenario_batch(100000 + i))
            xa = q.decode(q.encode(x))
            rel += float(((x - xa) ** 2).sum() / ((x - mean) ** 2).sum()) / 30
    out = {"steps": it, "losses": np.array(losses, np.float64), "lr": np.array(lrs, np.float64),
           "refine_iters": np.array(two_iter, np.int64), "avg_rel_err": rel}
    for k, v in init.items():
        out["init." + k] = v
    np.savez_compressed(os.path.join(HERE, "trainer_scenario_d256_b4.npz"), **out)
    print("steps", it, "avg_rel_err", rel, "losses[-1]", losses[-1])


if __name__ == "__main__":
    main()
