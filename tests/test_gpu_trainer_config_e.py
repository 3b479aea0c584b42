"""BASELINE.json config E at its own shape -- QuantizerTrainer.step (/root/reference/quantization/quantization.py:641-719) at
dim 512, 8 bytes per frame, batches of 4,096 frames -- against a trajectory of the REFERENCE trainer on the same seeded inputs
(tests/golden/make_golden_trainer.py config_e -> trainer_config_e_d512_b8.npz: 12 + 12 iterations on CPU).

Single process: same initial parameters, the learning rate EXACTLY at every step, the same refine-iteration draws, the losses of
step 0 (identical parameters) within 1e-4 and of every later step within 1e-2 (the two entropy diagnostics: 1e-3 absolute), every 37th row of the centers (after steps 1, 2, 13, 14, 15 and at
the end) and of the final classifier no further from the reference's than 1.5 x the reference's OWN feature-permuted run is.  Data parallel: two ranks on half batches (both on cuda:0, gloo: the test box has one GPU; on a node the same code
runs over RCCL) end with the parameters of the single process and follow the same reference trajectory."""
import os
import random
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from golden import gen

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FX_PATH = os.path.join(HERE, "golden", "trainer_config_e_d512_b8.npz")


def _check_against_reference(fx, losses, lrs, final):
    ref = fx["losses"]
    losses = np.asarray(losses)
    assert losses.shape == ref.shape, (losses.shape, ref.shape)
    assert np.array_equal(np.asarray(lrs), fx["lr"]), "learning-rate schedule differs from the reference's"
    rel = np.abs(losses - ref) / np.maximum(np.abs(ref), 1e-3)
    # step 0 sees identical parameters.  The fourth loss is the normalised index entropy (log K - H) / log K ~ 1.5e-3: a function of
    # the integer code counts, so ONE near-tie code of the 65,536 that differs from the reference's moves it by ~1e-3 of itself
    assert rel[0, :3].max() <= 1e-4 and rel[0, 3] <= 5e-3, rel[0]
    print("max relative deviation per loss over the trajectory:", rel.max(axis=0))
    assert rel[:, :2].max() <= 1e-2, (rel[:, :2].max(), np.unravel_index(rel[:, :2].argmax(), rel[:, :2].shape))
    # the two entropy diagnostics are (log K - H) / log K of values ~1e-3 .. 7e-3 (differences of near-equal numbers; the index
    # entropy moves with single codes): bounded absolutely
    assert np.abs(losses[:, 2:] - ref[:, 2:]).max() <= 1e-3, np.abs(losses[:, 2:] - ref[:, 2:]).max(axis=0)
    # Parameters.  Adam normalises every element's step to ~lr whatever the gradient's size, so a near-tie code that flips (a
    # handful per 4,096 frames at this state, in the reference's own feature-permuted run as well) re-directs whole rows: two runs
    # of the REFERENCE drift apart at the 1e-4 .. 1e-3 level within a few steps while their losses stay together.  The fixture holds
    # that drift (make_golden_trainer.py: perm_dev_*); this trainer may be no further from the reference than 1.5 x of it.
    for k in ("centers", "to_logits.weight"):
        want = fx["final_rows37." + k]
        got = final[k].reshape(-1, final[k].shape[-1])[::37]
        d = np.abs(got - want)
        yard = float(fx["perm_dev_mean.final." + k])
        print(k, "final rows: mean deviation %.3e (the reference's own permuted run: %.3e), max %.2e, share within 5e-3: %.4f (%.4f)"
              % (d.mean(), yard, d.max(), (d <= 5e-3).mean(), float(fx["perm_dev_share_within_5e-3.final." + k])))
        assert d.mean() <= 1.5 * yard + 2e-5, (k, d.mean(), yard)
        assert (d <= 5e-3).mean() >= float(fx["perm_dev_share_within_5e-3.final." + k]) - 0.05


def _check_rows_after(fx, it, centers):
    key = "centers_rows37_after_step%d" % it
    if key not in fx:
        return
    d = np.abs(centers.reshape(-1, centers.shape[-1])[::37] - fx[key])
    yard = float(fx["perm_dev_mean.centers_after_step%d" % it])
    print("centers after step %d: mean deviation %.3e (the reference's own permuted run: %.3e)" % (it, d.mean(), yard))
    assert d.mean() <= 1.5 * yard + 2e-5, (it, d.mean(), yard)


def test_config_e_trajectory_matches_reference():
    from quantization_amd import QuantizerTrainer
    fx = np.load(FX_PATH)
    D, B, P1, P2, seed = int(fx["D"]), int(fx["batch"]), int(fx["P1"]), int(fx["P2"]), int(fx["seed"])
    assert (D, int(fx["bytes"]), B) == (512, 8, 4096)
    torch.manual_seed(seed)
    random.seed(seed)
    dev = torch.device("cuda:0")
    tr = QuantizerTrainer(dim=D, bytes_per_frame=int(fx["bytes"]), device=dev, phase_one_iters=P1, phase_two_iters=P2)
    assert tr.fused_step
    assert np.array_equal(tr.quantizer.centers.detach().cpu().numpy(), fx["init.centers"])
    state = random.getstate()
    losses, lrs, shapes, it = [], [], [], 0
    while not tr.done():
        shapes.append((tr.quantizer.codebook_size, tr.quantizer.num_codebooks))
        lrs.append(tr.optim.param_groups[0]["lr"])
        tr.step(torch.from_numpy(gen.make_x(int(fx["data_seed"]) + it, B, D)).to(dev))
        losses.append(tr.last_losses)
        it += 1
        _check_rows_after(fx, it, tr.quantizer.centers.detach().cpu().numpy())
    assert it == int(fx["steps"]) == P1 + P2 + 1 and np.array_equal(np.array(shapes), fx["shapes"])
    random.setstate(state)                 # one draw per step (:651)
    assert np.array_equal(np.array([2 if random.random() < 0.5 else 1 for _ in range(it)]), fx["refine_iters"])
    final = {k: v.detach().cpu().numpy() for k, v in tr.get_quantizer().state_dict().items()}
    _check_against_reference(fx, losses, lrs, final)


def _run(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    from quantization_amd import QuantizerTrainer
    fx = np.load(FX_PATH)
    D, B, P1, P2, seed = int(fx["D"]), int(fx["batch"]), int(fx["P1"]), int(fx["P2"]), int(fx["seed"])
    dev = torch.device("cuda:0")
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(seed)      # (rank 0's initial parameters are broadcast; the seed of the reference run makes them the fixture's)
    random.seed(seed)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=int(fx["bytes"]), device=dev, phase_one_iters=P1, phase_two_iters=P2,
                          data_parallel=(world > 1))
    assert tr.fused_step
    it, losses, lrs = 0, [], []
    while not tr.done():
        x = torch.from_numpy(gen.make_x(int(fx["data_seed"]) + it, B, D))
        if world > 1:
            shard = B // world
            x = x[rank * shard:(rank + 1) * shard]
        lrs.append(tr.optim.param_groups[0]["lr"])
        tr.step(x.to(dev))
        losses.append(tr.last_losses)
        it += 1
    sd = {k: v.detach().cpu().numpy() for k, v in tr.get_quantizer().state_dict().items()}
    np.savez(out_path % rank, losses=np.array(losses), lrs=np.array(lrs), **sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_config_e_two_ranks_on_the_device():
    fx = np.load(FX_PATH)
    tmp = tempfile.mkdtemp()
    single, dp = os.path.join(tmp, "single_%d.npz"), os.path.join(tmp, "dp_%d.npz")
    mp.spawn(_run, args=(1, 0, single), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), dp), nprocs=2, join=True)
    a, r0, r1 = np.load(single % 0), np.load(dp % 0), np.load(dp % 1)
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.array_equal(r0[k], r1[k]), f"ranks diverged on {k}"
    assert np.array_equal(r0["losses"], r1["losses"])
    # Two ranks against one process.  The half-batch gradient sums are added in another order than one process adds them, so the
    # parameters differ in their last bits after the first step -- and from there the run is as chaotic as the reference's own: ONE
    # near-tie code that flips re-directs whole rows and, through the two scale factors, shifts every element by Adam's ~lr-sized
    # steps (tools/exp_perturb.py: a 1e-7 perturbation of the parameters moves 0 .. 1 of 4,096 codes per encode at this shape).
    # Until round 5 no code happened to flip within these 25 steps (96.8 % of the elements within 1e-5); with the shortlists in
    # position order (round 6) the tie-breaks fall differently and one flips at step 2.  So: the first steps agree to rounding,
    # and the end states are no further apart than 1.5 x what two runs of the REFERENCE are (its feature-permuted run,
    # perm_dev_mean.final.* of the fixture) -- the yardstick every trajectory comparison of this file uses.
    rel01 = np.abs(r0["losses"][:2] - a["losses"][:2]) / np.maximum(np.abs(a["losses"][:2]), 1e-3)
    assert rel01.max() <= 1e-5, rel01
    for k in ("centers", "to_logits.weight"):
        d = np.abs(r0[k] - a[k])
        yard = float(fx["perm_dev_mean.final." + k])
        print(k, "two ranks vs one process: mean deviation %.3e (the reference's own permuted run: %.3e)" % (d.mean(), yard))
        assert d.mean() <= 1.5 * yard + 2e-5, (k, d.mean(), yard)
    assert np.allclose(r0["losses"], a["losses"], rtol=1e-2, atol=1e-3), np.abs(r0["losses"] - a["losses"]).max()
    # and the two-rank run follows the reference's trajectory like the single process does
    _check_against_reference(fx, r0["losses"], r0["lrs"], {k: r0[k] for k in ("centers", "to_logits.weight")})
