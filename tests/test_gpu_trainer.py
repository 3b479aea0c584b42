"""QuantizerTrainer on the MI355X: the reference trainer's trajectory (CPU fixture) is followed
within fp32/near-tie tolerance, and the trained quantizer round-trips (the reference's own
integration scenario, test_quantization.py:11-48, at reduced size)."""
import os
import random

import numpy as np
import pytest
import torch

from golden import gen

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "trainer_d64_b4.npz"))


def test_trajectory_on_gpu_matches_reference():
    from quantization_amd import QuantizerTrainer
    D, B, P1, P2, seed = int(FX["D"]), int(FX["batch"]), int(FX["P1"]), int(FX["P2"]), int(FX["seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    dev = torch.device("cuda:0")
    tr = QuantizerTrainer(dim=D, bytes_per_frame=int(FX["bytes"]), device=dev, phase_one_iters=P1, phase_two_iters=P2)
    assert np.array_equal(tr.quantizer.centers.detach().cpu().numpy(), FX["init.centers"])
    losses, shapes, it = [], [], 0
    while not tr.done():
        shapes.append((tr.quantizer.codebook_size, tr.quantizer.num_codebooks))
        tr.step(torch.from_numpy(gen.make_x(9000 + it, B, D)).to(dev))
        losses.append(tr.last_losses)
        it += 1
    assert it == int(FX["steps"]) and np.array_equal(np.array(shapes), FX["shapes"])
    losses, ref = np.array(losses), FX["losses"]
    assert np.allclose(losses[0], ref[0], rtol=1e-4, atol=1e-5)
    assert np.allclose(losses, ref, rtol=1e-2, atol=1e-3), np.abs(losses - ref).max()
    q = tr.get_quantizer()
    for k in ("centers", "to_logits.weight"):
        a, b = q.state_dict()[k].cpu().numpy(), FX["final." + k]
        assert np.abs(a - b).max() <= 5e-3 * max(1.0, np.abs(b).max()), k


def test_train_then_roundtrip_and_checkpoint():
    from quantization_amd import Quantizer, QuantizerTrainer
    torch.manual_seed(1)
    random.seed(1)
    dev = torch.device("cuda:0")
    D = 64
    tr = QuantizerTrainer(dim=D, bytes_per_frame=4, device=dev, phase_one_iters=150, phase_two_iters=150)
    it = 0
    while not tr.done():
        tr.step(torch.from_numpy(gen.make_x(100 + it, 512, D)).to(dev))
        it += 1
    q = tr.get_quantizer()
    x = torch.from_numpy(gen.make_x(999, 4096, D)).to(dev)
    with torch.no_grad():
        codes = q.encode(x)
        assert codes.dtype == torch.uint8 and tuple(codes.shape) == (4096, 4)
        rel = float(((q.decode(codes) - x) ** 2).sum() / (x ** 2).sum())
        rel0 = float(((q.decode(q.encode(x, 0)) - x) ** 2).sum() / (x ** 2).sum())
    assert rel < 0.75 and rel <= rel0 + 1e-6, (rel, rel0)     # it learned something; refinement helps
    # state_dict round trip keeps the id (the reference suite's only assertion, test_train_hdf5.py:54)
    q2 = Quantizer(D, q.codebook_size, q.num_codebooks).to(dev)
    q2.load_state_dict(q.state_dict())
    assert q2.get_id() == q.get_id()
    with torch.no_grad():
        assert torch.equal(q2.encode(x), codes)
