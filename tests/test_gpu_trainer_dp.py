"""Data-parallel trainer on the HIP device, fused step (no autograd): two ranks on half batches end with the
parameters one process computes on the whole batch.  Both ranks share cuda:0 (the test box has one GPU), so the
process group is gloo on device tensors; on a multi-GPU node the same code runs over RCCL."""
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
D, BYTES, BATCH, P1, P2, SEED = 64, 4, 512, 4, 4, 13


def _run(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(SEED + rank)
    random.seed(SEED)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=dev, phase_one_iters=P1, phase_two_iters=P2,
                          data_parallel=(world > 1))
    assert tr.fused_step
    it, losses = 0, []
    while not tr.done():
        x = torch.from_numpy(gen.make_x(7000 + it, BATCH, D))
        if world > 1:
            shard = BATCH // world
            x = x[rank * shard:(rank + 1) * shard]
        tr.step(x.to(dev))
        losses.append(tr.last_losses)
        it += 1
    sd = {k: v.detach().cpu().numpy() for k, v in tr.get_quantizer().state_dict().items()}
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


def test_two_ranks_on_the_device_equal_single_process():
    tmp = tempfile.mkdtemp()
    single, dp = os.path.join(tmp, "single_%d.npz"), os.path.join(tmp, "dp_%d.npz")
    mp.spawn(_run, args=(1, 0, single), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), dp), nprocs=2, join=True)
    a, r0, r1 = np.load(single % 0), np.load(dp % 0), np.load(dp % 1)
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.array_equal(r0[k], r1[k]), f"ranks diverged on {k}"
        assert np.abs(r0[k] - a[k]).max() <= 2e-4 * max(1e-3, np.abs(a[k]).max()), (k, np.abs(r0[k] - a[k]).max())
    assert np.array_equal(r0["id_buf"], r1["id_buf"])
    assert np.allclose(r0["losses"], a["losses"], rtol=2e-4, atol=2e-5)
    assert np.array_equal(r0["losses"], r1["losses"])


def _run_rccl_single(rank, port, out_path, overlap):
    """One rank, backend nccl (= RCCL on ROCm): with `force_collectives` the step issues its collectives in a group of one
    (sums over one rank are identities), so the call pattern the 8-GPU run uses -- broadcast of the initial parameters,
    the forward all-reduce of the batch sums, the asynchronous all-reduce of the centers' slice of the gradient bucket
    overlapping the classifier's backward, the rest of the bucket behind it -- runs through RCCL on the 1-GPU test box."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MCQ_TRAINER_OVERLAP"] = "1" if overlap else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from quantization_amd import QuantizerTrainer
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    use_group = port != 0
    if use_group:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("nccl", rank=0, world_size=1)
    torch.manual_seed(SEED)
    random.seed(SEED)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=BYTES, device=dev, phase_one_iters=P1, phase_two_iters=P2,
                          data_parallel=use_group, force_collectives=use_group)
    assert tr.fused_step and tr._collective() == use_group
    it, losses = 0, []
    while not tr.done():
        tr.step(torch.from_numpy(gen.make_x(7000 + it, BATCH, D)).to(dev))
        losses.append(tr.last_losses)
        it += 1
    sd = {k: v.detach().cpu().numpy() for k, v in tr.get_quantizer().state_dict().items()}
    np.savez(out_path, losses=np.array(losses), **sd)
    if use_group:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_the_collectives_of_a_step_through_rccl(overlap):
    tmp = tempfile.mkdtemp()
    plain, coll = os.path.join(tmp, "plain.npz"), os.path.join(tmp, "rccl.npz")
    mp.spawn(_run_rccl_single, args=(0, plain, overlap), nprocs=1, join=True)
    mp.spawn(_run_rccl_single, args=(_free_port(), coll, overlap), nprocs=1, join=True)
    a, b = np.load(plain), np.load(coll)
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.array_equal(a[k], b[k]), f"{k} differs after the RCCL collectives of a one-rank group"
    assert np.array_equal(a["losses"], b["losses"])
