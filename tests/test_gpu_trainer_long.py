"""Full-length training runs on the MI355X ("missing" 2): the reference's dim-512 scenario at its default
length, and BASELINE config E (QuantizerTrainer.step, 20k iterations, dim 512, 8 bytes per frame, batches of 4,096).

1. test_quantization.py:87-129 (`_test_quantizer_trainer_double`): dim 512 frames made of TWO independent draws of the dim-256
   "MLP of noise" distribution, 8 bytes per frame, the default 10,000 + 10,000 iterations of 600 frames.  Its author built it to
   be compared with the dim-256 / 4-byte scenario (`:11-48`: same bytes per dimension, same distribution): the average relative
   reconstruction error of the two must agree.  Both run here at the same length.
2. Config E at its length: Gaussian frames, so the Shannon bound 2^(-2 rate) is a floor nothing can beat, and the reference
   trainer's own result after a SHORT schedule (400 + 400, tests/golden/make_golden_trainer_long.py) is a ceiling a run 25 times
   as long must stay below; the reconstruction loss must fall window by window inside each phase.
Also the end-of-training semantics (`done()` is `>`: one step more than the two phases, `:634`; `get_quantizer()` asserts `>=`,
`:741`)."""
import os
import random
import time

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _scenario_model(dev):
    dim = 256
    return nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Linear(dim, dim), nn.ReLU(), nn.LayerNorm(dim), nn.Linear(dim, dim)).to(dev)


def _avg_rel_err(q, gen_x, k=30):
    mean = q.get_data_mean()
    err = 0.0
    with torch.no_grad():
        for _ in range(k):                                       # test_quantization.py:41-46 / :120-125
            x = gen_x()
            err += float(((x - q.decode(q.encode(x))) ** 2).sum() / ((x - mean) ** 2).sum()) / k
    return err


def _train(tr, gen_x):
    steps = 0
    with torch.no_grad():
        while not tr.done():
            tr.step(gen_x())
            steps += 1
    return steps


def test_reference_scenario_dim512_doubled_at_default_length():
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")
    B = 600
    errs = {}
    for dim, nbytes in ((512, 8), (256, 4)):
        torch.manual_seed(1)
        random.seed(1)
        model = _scenario_model(dev)
        tr = QuantizerTrainer(dim=dim, bytes_per_frame=nbytes, device=dev)      # defaults: 10,000 + 10,000 (:581-583)
        assert tr.phase_one_iters == 10000 and tr.phase_two_iters == 10000

        def gen_x():
            with torch.no_grad():
                parts = []
                for _ in range(dim // 256):
                    x = torch.randn(B, 256, device=dev)
                    parts.append(model(x) + 0.05 * x)
                return torch.cat(parts, dim=1)

        t0 = time.perf_counter()
        with pytest.raises(AssertionError):
            tr.get_quantizer()                                   # (:741) not before the two phases are through
        steps = _train(tr, gen_x)
        torch.cuda.synchronize()
        assert steps == 20001 and tr.cur_iter == 20001           # done() is `>` (:634): one step beyond the two phases
        q = tr.get_quantizer()
        assert q.codebook_size == 256 and q.num_codebooks == nbytes
        errs[dim] = _avg_rel_err(q, gen_x)
        print(f"dim {dim}, {nbytes} bytes, 20,001 steps of {B} frames in {time.perf_counter() - t0:.1f} s: "
              f"avg relative error {errs[dim]:.5f}")
    # the doubled problem at twice the bytes: the same error per dimension (two independent halves could simply be coded apart;
    # a joint code can only be as good or a little better)
    assert abs(errs[512] - errs[256]) <= 0.03 * errs[256], errs


def test_config_e_at_its_length():
    from quantization_amd import QuantizerTrainer
    fx = np.load(os.path.join(HERE, "golden", "trainer_long_d512_b8.npz"))
    dim, nbytes, B = 512, 8, 4096
    assert int(fx["dim"]) == dim and int(fx["bytes"]) == nbytes
    shannon = 2.0 ** (-2.0 * nbytes * 8 / dim)
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    random.seed(5)
    tr = QuantizerTrainer(dim=dim, bytes_per_frame=nbytes, device=dev)          # 10,000 + 10,000
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    rec = []
    t0 = time.perf_counter()
    while not tr.done():
        tr.step(torch.randn(B, dim, device=dev, generator=g))
        rec.append(tr._last_losses[0])                            # device scalars: no sync per step
    torch.cuda.synchronize()
    total_s = time.perf_counter() - t0
    rec = torch.stack(rec).float().cpu().numpy()
    assert len(rec) == 20001
    q = tr.get_quantizer()
    err = _avg_rel_err(q, lambda: torch.randn(600, dim, device=dev, generator=g))
    print(f"config E: 20,001 steps of {B} frames in {total_s:.1f} s ({total_s / 20001 * 1e3:.3f} ms per step); relative error "
          f"{err:.5f}; reference after 400 + 400: {float(fx['avg_rel_err']):.5f}; Shannon bound {shannon:.5f}")
    # windows of 1,000 steps: the reconstruction loss falls (to 0.2 % of noise) inside each phase
    w = rec[:20000].reshape(20, 1000).mean(axis=1)
    for ph in (w[:10], w[10:]):
        assert (np.diff(ph) <= 2e-3 * ph[:-1]).all(), w
    assert w[10] < w[9] or w[19] < w[9], w                        # the second phase (8 x 256) ends below the first (16 x 16)
    assert shannon <= err <= float(fx["avg_rel_err"]), (shannon, err, float(fx["avg_rel_err"]))
    assert err <= 1.08 * shannon, (err, shannon)
