"""JointCodebookLoss (reference: quantization/prediction.py): drop-in surface on CPU, values and gradients against
fixtures captured from the reference (tests/golden/make_golden_jcl.py) on the GPU."""
import glob
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "jcl_*.npz")))


def _module(fx, checkpoint):
    from quantization_amd import JointCodebookLoss
    m = JointCodebookLoss(predictor_channels=int(fx["pc"]), num_codebooks=int(fx["ncb"]), hidden_channels=int(fx["hidden"]),
                          codebook_size=int(fx["K"]), reduction=str(fx["reduction"]), checkpoint=checkpoint)
    sd = {k[len("state."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("state.")}
    assert set(sd) == set(m.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in m.state_dict().items())
    m.load_state_dict(sd)       # a reference checkpoint loads unchanged
    return m


def test_surface_and_no_cpu_fallback():
    assert len(FIXTURES) == 3
    from quantization_amd import _lib
    fx = np.load(FIXTURES[0])
    m = _module(fx, True)
    with pytest.raises(_lib.McqError):
        m(torch.from_numpy(fx["predictor"]), torch.from_numpy(fx["indexes"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
@pytest.mark.parametrize("checkpoint", [True, False])
def test_loss_and_gradients_match_the_reference(path, checkpoint):
    fx = np.load(path)
    m = _module(fx, checkpoint).cuda()
    pred = torch.from_numpy(fx["predictor"]).cuda().requires_grad_(True)
    idx = torch.from_numpy(fx["indexes"]).cuda()
    loss = m(pred, idx)
    loss.backward()
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"])), (float(loss.detach()), float(fx["loss"]))
    got = {"grad_predictor": pred.grad}
    got.update({"grad." + k: p.grad for k, p in m.named_parameters()})
    for k, g in got.items():
        ref = fx[k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(g.cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale, (k, err, scale)
    # bit-reproducible (fixed-order scatter and reductions)
    m.zero_grad()
    pred2 = pred.detach().clone().requires_grad_(True)
    loss2 = m(pred2, idx)
    loss2.backward()
    assert torch.equal(loss2.detach(), loss.detach()) and torch.equal(pred2.grad, pred.grad)


@pytest.mark.gpu
def test_uint8_codes_from_encode_feed_the_loss():
    """the step after encode in the reference's integration script (test_train_hdf5.py:120-121)"""
    from quantization_amd import JointCodebookLoss, Quantizer
    torch.manual_seed(0)
    q = Quantizer(64, 256, 4).cuda()
    x = torch.randn(300, 64, device="cuda")
    with torch.no_grad():
        codes = q.encode(x)                       # uint8 (300, 4)
    m = JointCodebookLoss(predictor_channels=64, num_codebooks=4, hidden_channels=128).cuda()
    loss = m(x, codes)
    loss.backward()
    assert torch.isfinite(loss) and float(loss.detach()) > 0 and all(p.grad is not None for p in m.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("B,P,N,H,K", [(1, 8, 2, 16, 16), (3, 20, 3, 70, 32), (130, 33, 5, 96, 128)])
def test_ragged_sizes_against_the_torch_op_sequence(B, P, N, H, K):
    """sizes the fixtures do not hold (single frame, odd channel counts, codebook counts that are not powers of two)
    against the reference's op sequence (prediction.py:38-81) written with torch ops on the same device"""
    import torch.nn.functional as F
    from quantization_amd import JointCodebookLoss
    torch.manual_seed(B)
    m = JointCodebookLoss(P, N, H, K, reduction="mean", checkpoint=False).cuda()
    pred = torch.randn(B, P, device="cuda", requires_grad=True)
    idx = torch.randint(0, K, (B, N), device="cuda")
    if B > 2:
        idx[1] = -100

    def ref():
        first = idx[:, :-1].clamp(min=0) + torch.arange(0, (N - 1) * K, step=K, device=idx.device)
        emb = F.embedding(first, m.codebook_embedding.weight) * (0.5 * ((H / N) ** 0.5))
        a = torch.relu(torch.cumsum(torch.cat((m.linear1(pred).unsqueeze(1), emb), dim=1), dim=1))
        lp = torch.matmul(a.transpose(0, 1), m.linear2_weight.transpose(1, 2)).transpose(0, 1)
        lp = lp + torch.matmul(pred, m.linear2b_weight.transpose(1, 2)).transpose(0, 1) + m.linear2_bias
        return F.cross_entropy(lp.reshape(-1, K), idx.reshape(-1), ignore_index=-100, reduction="mean")

    grads = []
    for fn in (lambda: m(pred, idx), ref):
        m.zero_grad()
        pred.grad = None
        loss = fn()
        loss.backward()
        grads.append((loss.detach(), pred.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    (la, ga, pa), (lb, gb, pb) = grads
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max() + 1e-9)
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-4 * float(pb[k].abs().max() + 1e-9), k


def _torch_rows(m, pred, idx):
    """the reference's op sequence (prediction.py:38-82) with reduction='none', in torch"""
    import torch.nn.functional as F
    N, K, H = m.num_codebooks, m.codebook_size, m.hidden_channels
    p2 = pred.reshape(-1, pred.shape[-1])
    i2 = idx.reshape(-1, N).to(torch.int64)
    first = i2[:, :-1].clamp(min=0) + torch.arange(0, (N - 1) * K, K, device=i2.device)
    emb = F.embedding(first, m.codebook_embedding.weight) * (0.5 * ((H / N) ** 0.5))
    hp = F.linear(p2, m.linear1.weight, m.linear1.bias)
    a = torch.relu(torch.cumsum(torch.cat((hp.unsqueeze(1), emb), dim=1), dim=1))
    z = torch.matmul(a.transpose(0, 1), m.linear2_weight.transpose(1, 2)).transpose(0, 1)
    z = z + torch.matmul(p2, m.linear2b_weight.transpose(1, 2)).transpose(0, 1) + m.linear2_bias
    return F.cross_entropy(z.reshape(-1, K), i2.reshape(-1), ignore_index=-100, reduction="none")


@pytest.mark.gpu
def test_reduction_none_and_autocast():
    """reduction='none' gives the per-(frame, codebook) losses in the reference's order with the right gradient; under
    torch.autocast the fp32 kernels still see fp32 tensors (same loss and gradients as without it)."""
    fx = np.load(FIXTURES[0])
    m = _module(fx, False).cuda()
    pred = torch.from_numpy(fx["predictor"]).cuda()
    idx = torch.from_numpy(fx["indexes"]).cuda()
    m.reduction = "none"
    p1 = pred.clone().requires_grad_(True)
    rows = m(p1, idx)
    ref_rows = _torch_rows(m, pred, idx)
    assert rows.shape == ref_rows.shape
    assert torch.allclose(rows, ref_rows, rtol=1e-4, atol=1e-4)
    w = torch.rand_like(rows)
    (rows * w).sum().backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    p2 = pred.clone().requires_grad_(True)
    (_torch_rows(m, p2, idx) * w).sum().backward()
    assert torch.allclose(p1.grad, p2.grad, rtol=1e-3, atol=1e-4 * float(p2.grad.abs().max()))
    for k, p in m.named_parameters():
        assert torch.allclose(g1[k], p.grad, rtol=1e-3, atol=2e-4 * float(p.grad.abs().max())), k
    # autocast
    m.reduction = "sum"
    m.zero_grad()
    p3 = pred.clone().requires_grad_(True)
    plain = m(p3, idx)
    plain.backward()
    gp = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    p4 = pred.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        ac = m(p4, idx)
    ac.backward()
    assert ac.dtype == torch.float32 and torch.equal(ac.detach(), plain.detach()) and torch.equal(p3.grad, p4.grad)
    assert all(torch.equal(gp[k], p.grad) for k, p in m.named_parameters())
    # the quantizer's own loss under autocast
    from quantization_amd import Quantizer
    torch.manual_seed(0)
    q = Quantizer(64, 256, 4).cuda()
    x = torch.randn(300, 64, device="cuda")
    la = q.compute_loss(x, 1)
    sum(la[:3]).backward()
    ga = {k: p.grad.clone() for k, p in q.named_parameters()}
    q.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        lb = q.compute_loss(x, 1)
        tot = sum(lb[:3])
    tot.backward()
    assert all(torch.equal(a.detach(), b.detach()) for a, b in zip(la, lb))
    assert all(torch.equal(ga[k], p.grad) for k, p in q.named_parameters())


@pytest.mark.gpu
def test_downstream_scenario_follows_the_reference():
    """The reference's downstream scenario (test_train_hdf5.py:79-134) at a size the CPU reference trains in seconds: a Quantizer
    TRAINED BY THE REFERENCE encodes seeded frames on the MI355X, a JointCodebookLoss learns to predict the codes from the frames
    (Adam, lr 1e-3, StepLR as in that script).  tests/golden/make_golden_downstream.py ran the same 300 steps with the reference
    on CPU from the same initial predictor: the loss curves must coincide -- step 0 (identical parameters, the codes of 512
    frames) to 1e-5, every step to 1 %, the last 20 steps' mean to 0.3 %."""
    from quantization_amd import JointCodebookLoss, Quantizer
    from golden import gen
    fx = np.load(os.path.join(HERE, "golden", "downstream_d64_b4.npz"))
    D, NB, B, steps = int(fx["D"]), int(fx["bytes"]), int(fx["B"]), int(fx["steps"])
    dev = torch.device("cuda:0")
    q = Quantizer(D, 256, NB)
    q.load_state_dict({k[len("quantizer."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("quantizer.")})
    q = q.to(dev)
    predictor = JointCodebookLoss(predictor_channels=D, num_codebooks=NB)
    predictor.load_state_dict({k[len("predictor_init."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("predictor_init.")})
    predictor = predictor.to(dev)
    optim = torch.optim.Adam(predictor.parameters(), lr=0.001, betas=(0.9, 0.98), eps=1e-9, weight_decay=1.0e-06)
    scheduler = torch.optim.lr_scheduler.StepLR(optim, step_size=2000, gamma=0.5)
    losses = []
    for s in range(steps):
        x = torch.from_numpy(gen.make_x(int(fx["data_seed"]) + s, B, D)).to(dev)
        with torch.no_grad():
            encoding = q.encode(x)
        assert encoding.dtype == torch.uint8 and tuple(encoding.shape) == (B, NB)
        loss = predictor(x, encoding) / x.shape[0]
        losses.append(float(loss.detach()))
        loss.backward()
        optim.step()
        optim.zero_grad()
        scheduler.step()
    losses, ref = np.array(losses), fx["losses"]
    rel = np.abs(losses - ref) / ref
    print("downstream scenario: first %.4f (reference %.4f), last-20 mean %.4f (%.4f), largest relative deviation %.2e"
          % (losses[0], ref[0], losses[-20:].mean(), ref[-20:].mean(), rel.max()))
    assert rel[0] <= 1e-5 and rel.max() <= 1e-2, (rel[0], rel.max())
    assert abs(losses[-20:].mean() - ref[-20:].mean()) <= 3e-3 * ref[-20:].mean()
    assert losses[-20:].mean() < 0.45 * losses[0]                   # it learned to predict the codes
