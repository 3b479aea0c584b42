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


def run(perm):
    """One run of the reference trainer.  perm: None, or a permutation of the feature axis applied to the initial parameters and to
    every batch -- mathematically the same run with a different fp32 summation order inside torch's GEMMs (the reference's own
    reorder noise; rows are returned in the original feature order)."""
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = refq.QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1,
                               phase_two_iters=P2)
    init = {k: v.detach().numpy().copy() for k, v in tr.quantizer.state_dict().items()}
    pm = np.arange(D) if perm is None else perm
    inv = np.argsort(pm)
    if perm is not None:
        with torch.no_grad():
            tr.quantizer.centers.copy_(tr.quantizer.centers[:, :, pm].clone())
            tr.quantizer.to_logits.weight.copy_(tr.quantizer.to_logits.weight[:, pm].clone())
    losses, lrs, shapes, two_iter = [], [], [], []
    rows_after = {}
    scal = []
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
        tr.step(torch.from_numpy(np.ascontiguousarray(gen.make_x(DATA_SEED + it, BATCH, D)[:, pm])))
        iters, vals = rec["calls"][0]          # the training call is the first compute_loss of the step
        two_iter.append(iters)
        losses.append(vals)
        it += 1
        print("step", it, vals, flush=True)
        scal.append((float(tr.quantizer.centers_scale), float(tr.quantizer.logits_scale)))
        if not KEEP_FINAL and it in (1, 2, 13, 14, 15):      # every 37th row of the centers after this many steps
            c = tr.quantizer.centers.detach().numpy()[:, :, inv]
            rows_after[it] = c.reshape(-1, c.shape[-1])[::37].copy()
    refq.Quantizer.compute_loss = orig
    final = {k: v.detach().numpy().copy() for k, v in tr.get_quantizer().state_dict().items()}
    final["centers"] = final["centers"][:, :, inv].copy()
    final["to_logits.weight"] = final["to_logits.weight"][:, inv].copy()
    return dict(init=init, losses=np.array(losses, np.float64), lrs=lrs, shapes=shapes, two_iter=two_iter, rows_after=rows_after,
                scal=scal, final=final, steps=it)


def rows37(v):
    return v.reshape(-1, v.shape[-1])[::37].copy() if v.ndim >= 2 else v[::37].copy()


def main():
    torch.set_num_threads(8)
    r = run(None)
    it, losses, shapes, lrs, final = r["steps"], r["losses"], r["shapes"], r["lrs"], r["final"]
    out = {"D": D, "bytes": BYTES, "batch": BATCH, "P1": P1, "P2": P2, "seed": SEED, "steps": it,
           "losses": losses, "lr": np.array(lrs, np.float64),
           "shapes": np.array(shapes, np.int64), "refine_iters": np.array(r["two_iter"], np.int64)}
    for k, v in r["init"].items():
        out["init." + k] = v
    out["data_seed"] = DATA_SEED
    out["scales_after_step"] = np.array(r["scal"], np.float64)      # (centers_scale, logits_scale) after every step
    for k, v in r["rows_after"].items():
        out["centers_rows37_after_step%d" % k] = v
    for k, v in final.items():
        if KEEP_FINAL or v.ndim == 0:
            out["final." + k] = v
        elif k != "id_buf":
            out["final_rows37." + k] = rows37(v)
    if not KEEP_FINAL:
        # The reference's OWN reorder noise along this trajectory: the same run with the feature axis permuted.  Adam normalises
        # every element's step to ~lr, so a near-tie code that flips (a handful per 4,096 frames at this state) re-directs whole
        # rows: parameters of two runs of the reference drift apart at the 1e-4 .. 1e-3 level within a few steps while the losses
        # stay together.  The test bounds the HIP trainer's distance from the reference by this yardstick.
        p = run(np.random.RandomState(5).permutation(D))
        assert p["two_iter"] == r["two_iter"] and p["lrs"] == lrs
        for k, v in p["rows_after"].items():
            out["perm_dev_mean.centers_after_step%d" % k] = float(np.abs(v - r["rows_after"][k]).mean())
        for k in ("centers", "to_logits.weight"):
            out["perm_dev_mean.final." + k] = float(np.abs(rows37(p["final"][k]) - rows37(final[k])).mean())
            out["perm_dev_share_within_5e-3.final." + k] = float((np.abs(rows37(p["final"][k]) - rows37(final[k])) <= 5e-3).mean())
        out["perm_losses"] = p["losses"]
        print({k: v for k, v in out.items() if k.startswith("perm_dev")})
        print("largest relative loss deviation of the permuted run:", (np.abs(p["losses"] - losses) / np.maximum(np.abs(losses), 1e-3)).max(axis=0))
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), **out)
    print("steps", it, "shapes", shapes, "lr", lrs)
    print("losses[0]", losses[0], "losses[-1]", losses[-1])


if __name__ == "__main__":
    main()
