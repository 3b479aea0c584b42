"""Where does the config-E trajectory leave the reference's?  Every 37th row of the centers after steps 1, 2, 13, 14, 15 against
tests/golden/trainer_config_e_d512_b8.npz (the reference's rows at the same points)."""
import os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import gen
from quantization_amd import QuantizerTrainer
fx = np.load(os.path.join(ROOT, "tests", "golden", "trainer_config_e_d512_b8.npz"))
D, B, P1, P2, seed = int(fx["D"]), int(fx["batch"]), int(fx["P1"]), int(fx["P2"]), int(fx["seed"])
torch.manual_seed(seed); random.seed(seed)
dev = torch.device("cuda:0")
tr = QuantizerTrainer(dim=D, bytes_per_frame=int(fx["bytes"]), device=dev, phase_one_iters=P1, phase_two_iters=P2)
it = 0
while not tr.done():
    tr.step(torch.from_numpy(gen.make_x(int(fx["data_seed"]) + it, B, D)).to(dev))
    it += 1
    print(f"step {it}: losses {np.array(tr.last_losses)} ref {fx['losses'][it-1]}  scales {float(tr.quantizer.centers_scale):.7f} {float(tr.quantizer.logits_scale):.7f} ref {fx['scales_after_step'][it-1]}")
    key = "centers_rows37_after_step%d" % it
    if key in fx:
        c = tr.quantizer.centers.detach().cpu().numpy()
        got, want = c.reshape(-1, c.shape[-1])[::37], fx[key]
        d = np.abs(got - want)
        moved = np.abs(want - (prev if it not in (1, 14) else want)).mean() if False else 0
        print(f"after step {it}: shape {got.shape} max {d.max():.3e} mean {d.mean():.3e} share<=1e-4 {(d<=1e-4).mean():.4f} |want| mean {np.abs(want).mean():.3e}")
        rows = d.max(axis=1)
        print("   rows with max dev > 1e-3:", int((rows > 1e-3).sum()), "of", len(rows), " per-row max:", np.sort(rows)[-5:])
