"""Data-parallel trainer: two gloo ranks on half batches must end with the parameters a
single process computes on the whole batch (one gradient all-reduce + one forward all-reduce
of the batch sums per step).  CPU-only; the index search is injected from the oracle."""
import os
import random
import socket
import sys
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

from golden import gen

D, BYTES, BATCH, P1, P2, SEED = 32, 2, 128, 3, 3, 11


def _data(it):
    return torch.from_numpy(gen.make_x(4000 + it, BATCH, D))


def _run(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import oracle_kernels
    from quantization_amd import QuantizerTrainer
    torch.set_num_threads(2)
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(SEED + rank)        # different init per rank: the trainer must broadcast rank 0's
    random.seed(SEED)                     # ranks share the Python RNG seed (refine-iteration draw, :651)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=torch.device("cpu"), phase_one_iters=P1,
                          phase_two_iters=P2, data_parallel=(world > 1))
    it = 0
    losses = []
    with oracle_kernels():
        while not tr.done():
            x = _data(it)
            if world > 1:
                shard = BATCH // world
                x = x[rank * shard:(rank + 1) * shard]
            tr.step(x)
            losses.append(tr.last_losses)
            it += 1
    sd = {k: v.detach().numpy() for k, v in tr.get_quantizer().state_dict().items()}
    np.savez(out_path % rank, losses=np.array(losses), **sd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_single_process():
    tmp = tempfile.mkdtemp()
    single = os.path.join(tmp, "single_%d.npz")
    # single process with rank 0's seed
    _run(0, 1, 0, single)
    dp = os.path.join(tmp, "dp_%d.npz")
    mp.spawn(_run, args=(2, _free_port(), dp), nprocs=2, join=True)
    a, r0, r1 = np.load(single % 0), np.load(dp % 0), np.load(dp % 1)
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.array_equal(r0[k], r1[k]), f"ranks diverged on {k}"
        assert np.allclose(r0[k], a[k], rtol=1e-4, atol=2e-5), (k, np.abs(r0[k] - a[k]).max())
    assert np.array_equal(r0["id_buf"], r1["id_buf"])
    # the reported losses are those of the whole batch on every rank
    assert np.allclose(r0["losses"], a["losses"], rtol=1e-4, atol=1e-5)
    assert np.allclose(r0["losses"], r1["losses"], rtol=0, atol=0)


def test_shard_plan_covers_batch():
    from quantization_amd.sharding import shard_bounds
    for B in (0, 1, 7, 8, 65536, 8 * 1048576 + 3):
        for W in (1, 2, 3, 8):
            cuts = [shard_bounds(B, W, r) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _run_sharded_encode(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from golden import fixtures
    from oracle_backend import oracle_kernels
    from quantization_amd import Quantizer
    from quantization_amd.sharding import encode_sharded, shard_bounds
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = fixtures.load("trained_d64_b4_p2")
    q = Quantizer(fx["D"], fx["K"], fx["N"])
    sd = q.state_dict()
    for k, v in fx["state"].items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd)
    x = torch.from_numpy(fx["x"][:301])          # 301 vectors: uneven shards
    with oracle_kernels(), torch.no_grad():
        local = encode_sharded(q, x, 2)                       # no collective on this path
        full = encode_sharded(q, x, 2, gather=True)           # optional all_gather of the codes
    lo, hi = shard_bounds(301, world, rank)
    np.savez(out_path % rank, local=local.numpy(), full=full.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_encode_two_ranks_matches_fixture():
    """Batch-sharded encode: each rank encodes its own shard (no data-path collective); the
    concatenation equals the single-process result, i.e. the reference fixture's codes."""
    from golden import fixtures
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "enc_%d.npz")
    mp.spawn(_run_sharded_encode, args=(2, _free_port(), path), nprocs=2, join=True)
    r0, r1 = np.load(path % 0), np.load(path % 1)
    fx = fixtures.load("trained_d64_b4_p2")
    want = fx["codes_it2"][:301]
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 151, 151, 301)
    cat = np.concatenate([r0["local"], r1["local"]])
    assert np.array_equal(r0["full"], r1["full"]) and np.array_equal(r0["full"], cat)
    bad = (cat != want).any(axis=1)
    assert (bad & (fx["margin_it2"][:301] >= fixtures.NEAR_TIE)).sum() == 0 and bad.sum() <= 1
