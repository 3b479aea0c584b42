"""QuantizerTrainer host logic (schedule, phase switch, optimiser wiring, loss) against the
trajectory captured from the reference trainer (tests/golden/make_golden_trainer.py).
CPU-only: the index search is injected from the oracle (tests/oracle_backend.py)."""
import os
import random

import numpy as np
import torch

from golden import gen
from oracle_backend import oracle_kernels

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "trainer_d64_b4.npz"))


def run_trainer():
    from quantization_amd import QuantizerTrainer
    D, B, P1, P2, seed = int(FX["D"]), int(FX["batch"]), int(FX["P1"]), int(FX["P2"]), int(FX["seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=int(FX["bytes"]), device=torch.device("cpu"), phase_one_iters=P1,
                          phase_two_iters=P2)
    init = {k: v.detach().numpy().copy() for k, v in tr.quantizer.state_dict().items()}
    losses, lrs, shapes = [], [], []
    it = 0
    with oracle_kernels():
        while not tr.done():
            shapes.append((tr.quantizer.codebook_size, tr.quantizer.num_codebooks))
            lrs.append(tr.optim.param_groups[0]["lr"])
            tr.step(torch.from_numpy(gen.make_x(9000 + it, B, D)))
            losses.append(tr.last_losses)
            it += 1
        q = tr.get_quantizer()
    return init, np.array(losses), np.array(lrs), np.array(shapes), it, q


def test_trajectory_matches_reference_trainer():
    init, losses, lrs, shapes, steps, q = run_trainer()
    # same torch seed => identical initial parameters (nn.Linear init, centers = copy of the weight)
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.array_equal(init[k], FX["init." + k]), k
    # done() uses '>' so P1 + P2 + 1 steps run; the switch to 256 x bytes happens after step P1
    assert steps == int(FX["steps"]) == int(FX["P1"]) + int(FX["P2"]) + 1
    assert np.array_equal(shapes, FX["shapes"])
    assert np.allclose(lrs, FX["lr"], rtol=0, atol=1e-12)            # Adam + StepLR(step=P/4, gamma .5), lr halved at the switch
    ref = FX["losses"]
    assert np.allclose(losses[0], ref[0], rtol=2e-5, atol=1e-6)     # first step: same ops on the same init
    assert np.allclose(losses, ref, rtol=5e-3, atol=5e-4), np.abs(losses - ref).max()
    sd = q.state_dict()
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        a, b = sd[k].detach().numpy(), FX["final." + k]
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max()), (k, np.abs(a - b).max())


def test_get_quantizer_asserts_before_the_end():
    from quantization_amd import QuantizerTrainer
    tr = QuantizerTrainer(dim=16, bytes_per_frame=1, device=torch.device("cpu"), phase_one_iters=2, phase_two_iters=2)
    try:
        tr.get_quantizer()
        raise RuntimeError("expected AssertionError")
    except AssertionError:
        pass
