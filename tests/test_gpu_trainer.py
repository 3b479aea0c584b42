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


def test_decode_backward_kernel_matches_autograd_and_is_deterministic():
    from quantization_amd import Quantizer
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    for (D, K, N, B) in [(64, 256, 4, 1000), (40, 16, 8, 333), (512, 256, 8, 4096)]:
        q = Quantizer(D, K, N).to(dev)
        with torch.no_grad():
            q.centers.normal_()
            q.centers_scale.fill_(0.05)
        idx = torch.randint(0, K, (B, N), device=dev)
        w = torch.randn(B, D, device=dev)
        grads = []
        for _ in range(2):
            q.zero_grad()
            (q.decode(idx) * w).sum().backward()
            grads.append((q.centers.grad.clone(), q.centers_scale.grad.clone()))
        assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])   # bit-reproducible
        # reference: the same function written with torch ops (quantization.py:131-148) under autograd
        centers = q.centers.detach().clone().requires_grad_(True)
        cscale = q.centers_scale.detach().clone().requires_grad_(True)
        c = (cscale * 10.0).exp() * centers
        y = torch.gather(c, 1, idx.t().contiguous().unsqueeze(-1).expand(N, B, D)).sum(dim=0)
        (y * w).sum().backward()
        assert torch.allclose(grads[0][0], centers.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(grads[0][1], cscale.grad, rtol=1e-3, atol=1e-2)


def _grads(q):
    return {n: p.grad.detach().clone() for n, p in q.named_parameters()}


@pytest.mark.parametrize("D,K,N,B", [(64, 256, 4, 1000), (40, 16, 8, 333), (96, 64, 2, 257), (512, 256, 8, 4096), (128, 16, 16, 2048),
                                     (32, 32, 4, 1), (32, 128, 2, 3), (48, 16, 64, 65)])
@pytest.mark.parametrize("iters", [0, 2])
def test_fused_loss_kernels_match_the_torch_op_formulation(D, K, N, B, iters):
    """compute_loss through mcq_logits_argmax / mcq_recon_fwd / mcq_loss_fwd / mcq_loss_bwd vs the reference's
    own torch op sequence under autograd (quantization.py:211-242; taken when x requires grad): the four
    losses to 2e-5 relative, every parameter gradient to 2e-4 of its largest entry."""
    from quantization_amd import Quantizer
    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    q = Quantizer(D, K, N).to(dev)
    with torch.no_grad():
        q.to_logits.bias.normal_(std=0.1)
        q.centers.mul_(3.0)
        q.logits_scale.fill_(0.03)
        q.centers_scale.fill_(-0.02)
    x = torch.randn(B, D, device=dev)
    w = [1.0, 1.0, 0.01]

    def total(losses):
        return losses[0] * w[0] + losses[1] * w[1] + losses[2] * w[2]

    q.zero_grad()
    lf = q.compute_loss(x, iters)
    total(lf).backward()
    gf = _grads(q)
    q.zero_grad()
    lt = q.compute_loss(x.clone().requires_grad_(True), iters)
    total(lt).backward()
    gt = _grads(q)
    for a, b, name in zip(lf, lt, ["recon", "logprob", "logits_entropy", "index_entropy"]):
        a, b = a.detach(), b.detach()
        assert abs(float(a) - float(b)) <= 2e-5 * max(1.0, abs(float(b))), (name, float(a), float(b))
    for n in gt:
        scale = float(gt[n].abs().max()) + 1e-12
        assert float((gf[n] - gt[n]).abs().max()) <= 2e-4 * scale, (n, float((gf[n] - gt[n]).abs().max()), scale)
    # bit-reproducible run to run (fixed-order reductions)
    q.zero_grad()
    lf2 = q.compute_loss(x, iters)
    total(lf2).backward()
    gf2 = _grads(q)
    assert all(torch.equal(a, b) for a, b in zip(lf, lf2)) and all(torch.equal(gf[n], gf2[n]) for n in gf)


def test_fused_step_equals_the_autograd_step():
    """QuantizerTrainer.step through _fused_loss_and_grads (forward kernels, mcq_loss_tail, backward kernels)
    against the same step through compute_loss + backward(): same losses and same parameters after steps in
    both phases (the phase switch included)."""
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")
    runs = []
    for fused in (True, False):
        torch.manual_seed(21)
        random.seed(21)
        tr = QuantizerTrainer(dim=96, bytes_per_frame=4, device=dev, phase_one_iters=6, phase_two_iters=6)
        tr.fused_step = fused
        losses = []
        for it in range(12):
            tr.step(torch.from_numpy(gen.make_x(500 + it, 768, 96)).to(dev))
            losses.append(tr.last_losses)
        runs.append((np.array(losses), {k: v.detach().cpu().numpy() for k, v in tr.quantizer.state_dict().items()},
                     tr.quantizer.codebook_size))
    (lf, pf, kf), (la, pa, ka) = runs
    assert kf == ka == 256
    assert np.allclose(lf, la, rtol=2e-4, atol=2e-5), np.abs(lf - la).max()
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.abs(pf[k] - pa[k]).max() <= 2e-4 * max(1e-3, np.abs(pa[k]).max()), (k, np.abs(pf[k] - pa[k]).max())


def test_decode_backward_on_uint8_codes_is_bit_identical():
    from quantization_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    for (D, K, N, B) in [(512, 256, 8, 4096), (40, 16, 16, 777), (100, 64, 2, 65)]:
        g = torch.randn(B, D, device=dev)
        idx = torch.randint(0, K, (B, N), device=dev)
        a = torch.empty(N, K, D, device=dev)
        b = torch.empty(N, K, D, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        assert L.mcq_decode_backward(g.data_ptr(), idx.data_ptr(), B, N, K, D, a.data_ptr(), st) == 0
        codes = idx.to(torch.uint8)
        assert L.mcq_decode_backward_u8(g.data_ptr(), codes.data_ptr(), B, N, K, D, b.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        ref = torch.zeros(N * K, D, device=dev, dtype=torch.float64)
        rows = (idx + torch.arange(N, device=dev) * K).reshape(-1)
        ref.index_add_(0, rows, g.double().unsqueeze(1).expand(-1, N, -1).reshape(-1, D))
        assert torch.allclose(a.double().reshape(N * K, D), ref, rtol=1e-5, atol=1e-4)


def test_adam_kernel_matches_torch_adam():
    """mcq_adam_step over a flat bucket against torch.optim.Adam (the reference's optimizer, quantization.py:722-727):
    parameters within 3e-8 after 25 steps with changing learning rates, padding gaps untouched."""
    from quantization_amd.trainer import _FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    shapes = [(8, 256, 24), (2048, 24), (2048,), (), ()]
    ref_params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in shapes]
    offs, off = [], 0
    for p in ref_params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_p = torch.zeros(off, device=dev)
    flat_g = torch.zeros(off, device=dev)
    mine = []
    for p, o in zip(ref_params, offs):
        q = torch.nn.Parameter(torch.empty(0, device=dev))
        q.data = flat_p[o:o + p.numel()].view(p.shape)
        q.data.copy_(p.data)
        q.grad = flat_g[o:o + p.numel()].view(p.shape)
        mine.append(q)
    kw = dict(lr=0.005, betas=(0.9, 0.98), eps=1e-9, weight_decay=1.0e-06)
    ref_opt = torch.optim.Adam(ref_params, **kw)
    my_opt = _FlatAdam(mine, flat_p, flat_g, **kw)
    sched_r = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=2.5, gamma=0.5)
    sched_m = torch.optim.lr_scheduler.StepLR(my_opt, step_size=2.5, gamma=0.5)
    for it in range(25):
        for p, q in zip(ref_params, mine):
            g = torch.randn(p.shape, device=dev) * (0.1 if it % 3 else 3.0)
            p.grad = g.clone()
            q.grad.copy_(g)
        ref_opt.step()
        my_opt.step()
        sched_r.step()
        sched_m.step()
        assert ref_opt.param_groups[0]["lr"] == my_opt.param_groups[0]["lr"]
    for p, q in zip(ref_params, mine):
        assert float((p.detach() - q.detach()).abs().max()) <= 3e-8, float((p.detach() - q.detach()).abs().max())


@pytest.mark.parametrize("B,N,K,D", [(4096, 8, 256, 512), (600, 4, 256, 256), (333, 8, 16, 40), (65, 2, 16, 30), (1000, 16, 16, 96),
                                     (4001, 4, 256, 256), (2500, 16, 256, 128), (2048, 8, 128, 1024)])
def test_weight_grad_kernel_matches_matmul(B, N, K, D):
    """mcq_weight_grad (fused scale / column sums) against the library formulation s * G^T x, G.sum(0): the fp32-MFMA kernel and
    -- the large tile-aligned shapes, with a ragged last stage among them -- the bf16-piece kernel (k_wgrad_bf3)."""
    from quantization_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(B + D)
    G = torch.randn(B, N * K, device=dev) * 0.01
    x = torch.randn(B, D, device=dev)
    s = torch.tensor([1.37], device=dev)
    gW = torch.empty(N * K, D, device=dev)
    gb = torch.empty(N * K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(L.mcq_weight_grad_workspace_bytes(B, N * K, D), dtype=torch.uint8, device=dev)
    assert L.mcq_weight_grad(G.data_ptr(), x.data_ptr(), B, N * K, D, s.data_ptr(), gW.data_ptr(), gb.data_ptr(), ws.data_ptr(),
                             ws.numel(), st) == 0
    ref = (G.double().t() @ x.double()) * 1.37
    assert float((gW.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    refb = G.double().sum(dim=0)
    assert float((gb.double() - refb).abs().max()) <= 2e-5 * float(refb.abs().max() + 1e-6)
    gW2 = torch.empty_like(gW)
    gb2 = torch.empty_like(gb)
    assert L.mcq_weight_grad(G.data_ptr(), x.data_ptr(), B, N * K, D, s.data_ptr(), gW2.data_ptr(), gb2.data_ptr(), ws.data_ptr(),
                             ws.numel(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(gW, gW2) and torch.equal(gb, gb2)        # fixed summation order


@pytest.mark.parametrize("D,K,N", [(64, 16, 8), (100, 256, 2), (512, 256, 8), (40, 32, 64 // 4)])
def test_data_mean_in_the_prepared_state(D, K, N):
    """get_data_mean() (quantization.py:67-75) of the scaled centers is formed by mcq_prepare inside the prepared blob
    (k_centers_mean): equal to centers.mean(dim=1).sum(dim=0) up to fp32 summation order"""
    from quantization_amd import Quantizer, _lib
    torch.manual_seed(D + K)
    q = Quantizer(D, K, N).cuda()
    with torch.no_grad():
        q.centers.copy_(torch.randn_like(q.centers))
        q.centers_scale.fill_(0.03)
        blob = q._prepared()
        L = _lib.lib()
        off = L.mcq_prepared_mean_offset(N, K, D)
        Dp = L.mcq_padded_dim(D)
        got = blob[off:off + 4 * Dp].view(torch.float32)[:D].cpu()
        want = q.get_centers().mean(dim=1).sum(dim=0).cpu()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (got - want).abs().max()


def test_evaluation_between_steps_keeps_the_scale_factors_current():
    """A no_grad encode / decode between two trainer steps (periodic evaluation) rebuilds the derived state with
    HOST-formed scale factors; the next fused step must not pair it with the device factors of the step BEFORE the
    optimizer update (they differ by exp(10 * lr) ~ 5 % after one Adam step).  The fused gradients are compared with
    autograd's on the same parameters."""
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(31)
    random.seed(31)
    D, B = 64, 512
    tr = QuantizerTrainer(dim=D, bytes_per_frame=4, device=dev, phase_one_iters=50, phase_two_iters=50)
    q = tr.quantizer
    for it in range(3):
        tr.step(torch.from_numpy(gen.make_x(800 + it, B, D)).to(dev))
        xe = torch.from_numpy(gen.make_x(900 + it, 256, D)).to(dev)
        with torch.no_grad():
            q.decode(q.encode(xe))                 # evaluation: host-flavour derived state for the new parameters
        assert q._prep.flavour == "host" and q._scales_dev is None
    x = torch.from_numpy(gen.make_x(850, B, D)).to(dev)
    tr._fused_loss_and_grads(x, 1)
    fused = {n: p.grad.detach().clone() for n, p in q.named_parameters()}
    tr.optim.zero_grad()
    losses = q.compute_loss(x.clone().requires_grad_(True), 1)      # the reference's torch op sequence under autograd
    (losses[0] + losses[1] + losses[2] * tr.entropy_scale).backward()
    for n, p in q.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((fused[n] - p.grad).abs().max()) <= 2e-4 * scale, (n, float((fused[n] - p.grad).abs().max()), scale)
    tr.optim.zero_grad()


def test_an_autograd_step_after_a_fused_step_starts_from_clean_gradients():
    """Fused steps leave their gradients in the bucket (the next fused step overwrites them); a step that goes through
    autograd accumulates into .grad, so the bucket must be cleared first."""
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")
    runs = []
    for pattern in ((True, False, True, False), (False, False, False, False)):
        torch.manual_seed(41)
        random.seed(41)
        tr = QuantizerTrainer(dim=64, bytes_per_frame=4, device=dev, phase_one_iters=50, phase_two_iters=50)
        for it, fused in enumerate(pattern):
            tr.fused_step = fused
            tr.step(torch.from_numpy(gen.make_x(600 + it, 512, 64)).to(dev))
        runs.append({k: v.detach().cpu().numpy() for k, v in tr.quantizer.state_dict().items()})
    a, b = runs
    for k in ("centers", "to_logits.weight", "to_logits.bias", "logits_scale", "centers_scale"):
        assert np.abs(a[k] - b[k]).max() <= 2e-4 * max(1e-3, np.abs(b[k]).max()), (k, np.abs(a[k] - b[k]).max())


def test_flat_adam_state_dict_round_trip():
    """The flat moments and the step count travel in state_dict(): a trainer resumed from it continues exactly."""
    from quantization_amd import QuantizerTrainer
    dev = torch.device("cuda:0")

    def make():
        torch.manual_seed(51)
        random.seed(51)
        return QuantizerTrainer(dim=64, bytes_per_frame=4, device=dev, phase_one_iters=50, phase_two_iters=50)

    a = make()
    for it in range(4):
        a.step(torch.from_numpy(gen.make_x(700 + it, 512, 64)).to(dev))
    sd_opt, sd_sched, sd_q = a.optim.state_dict(), a.scheduler.state_dict(), a.quantizer.state_dict()
    assert sd_opt["state"][0]["step"] == 4 and sd_opt["state"][0]["exp_avg"].numel() == a._flat[0].numel()
    b = make()
    with torch.no_grad():
        for (n, p), (_, pa) in zip(b.quantizer.named_parameters(), a.quantizer.named_parameters()):
            p.copy_(pa)
    b.quantizer.invalidate_cache()
    b.optim.load_state_dict(sd_opt)
    b.scheduler.load_state_dict(sd_sched)
    b.cur_iter = a.cur_iter
    assert b.optim.t == 4 and b.optim.exp_avg.data_ptr() != a.optim.exp_avg.data_ptr()
    rs = random.getstate()
    for tr in (a, b):
        random.setstate(rs)
        for it in range(3):
            tr.step(torch.from_numpy(gen.make_x(710 + it, 512, 64)).to(dev))
    for (n, pa), (_, pb) in zip(a.quantizer.named_parameters(), b.quantizer.named_parameters()):
        assert torch.equal(pa, pb), n
    del sd_q
