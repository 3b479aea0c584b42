"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
fixtures captured from the reference.  Run on the MI355X: pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

from golden import fixtures, gen
from oracle.oracle import OracleQuantizer

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    # like the reference, decode() is differentiable unless autograd is off (test_quantization.py uses no_grad)
    with torch.no_grad():
        yield

ALL = fixtures.names()


def load_quantizer(state, D, K, N, device="cuda:0"):
    from quantization_amd import Quantizer
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd)
    if getattr(state, "scales_exp", None) is not None:      # a fixture's state: the scale factors of the reference's run (fixtures.PinnedState)
        q.pin_scale_factors(*state.scales_exp)
    return q.to(device)


def oracle_of(state):
    return OracleQuantizer(state["centers"], float(state["centers_scale"]), state["to_logits.weight"],
                           state["to_logits.bias"], float(state["logits_scale"]), scales_exp=getattr(state, "scales_exp", None))


@pytest.mark.parametrize("name", ALL)
def test_codes_bit_exact_vs_oracle_and_reference(name):
    fx = fixtures.load(name)
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    o = oracle_of(fx["state"])
    x = torch.from_numpy(fx["x"]).cuda()
    for it in fx["iters"]:
        got = q.encode(x, it, as_bytes=False)
        assert got.dtype == torch.int64 and tuple(got.shape) == (fx["B"], fx["N"])
        got = got.cpu().numpy()
        want = o.compute_indexes(fx["x"], it)
        nbad = int((got != want).any(axis=1).sum())
        assert nbad == 0, f"{name} iters={it}: {nbad} vectors differ from the oracle"
        fixtures.check_codes(fx, it, got, f"{name} iters={it} (HIP vs reference fixture)")


@pytest.mark.parametrize("name", ALL)
def test_bytes_and_decode(name):
    fx = fixtures.load(name)
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    o = oracle_of(fx["state"])
    it = fx["iters"][-1]
    x = torch.from_numpy(fx["x"]).cuda()
    ref_codes = torch.from_numpy(fx[f"codes_it{it}"].astype(np.int64)).cuda()
    if fx["K"] <= 256:
        b = q.encode(x, it)  # as_bytes=True
        assert b.dtype == torch.uint8
        assert np.array_equal(b.cpu().numpy(), o.encode(fx["x"], it))
        ref_bytes_np = fx[f"bytes_it{it}"]
        ref_bytes = torch.from_numpy(ref_bytes_np).cuda()
    else:
        # codebooks of 512 / 1,024 entries: no byte form (quantization.py:271 asserts); the indexes are decoded as they are
        with pytest.raises(AssertionError):
            q.encode(x, it)
        ref_bytes_np, ref_bytes = fx[f"codes_it{it}"], ref_codes
    y = q.decode(ref_bytes)
    assert y.dtype == torch.float32 and tuple(y.shape) == (fx["B"], fx["D"])
    y = y.cpu().numpy()
    assert np.array_equal(y, o.decode(ref_bytes_np))                   # bit-exact vs the oracle
    assert np.array_equal(y, q.decode(ref_codes).cpu().numpy())        # packed == unpacked int64
    head = fx["decode_head"]
    scale = np.abs(head).max()
    assert np.abs(y[:16] - head).max() <= 1e-5 * scale                 # 1e-5 relative vs the reference
    assert np.allclose((y.astype(np.float64) ** 2).sum(axis=1), fx["decode_rowsumsq"], rtol=1e-5)
    # int32 codes and a leading batch shape, as decode accepts any integer dtype / (*, N)
    y2 = q.decode(ref_codes.to(torch.int32).reshape(-1, 2, fx["N"])[:4])
    assert tuple(y2.shape) == (4, 2, fx["D"])
    assert np.array_equal(y2.reshape(-1, fx["D"]).cpu().numpy(), y[:8])


def test_logits_bit_exact_vs_oracle():
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    o = oracle_of(fx["state"])
    x = fx["x"][:200]
    got = q.logits_kernel(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, o.logits(x))


def test_shapes_ragged_and_empty():
    fx = fixtures.load("synth_d40_k64_n8")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    o = oracle_of(fx["state"])
    x = fx["x"]
    for B in (1, 3, 63, 64, 65, 130):
        got = q.encode(torch.from_numpy(x[:B]).cuda(), 2).cpu().numpy()
        assert np.array_equal(got, o.encode(x[:B], 2)), B
    e = q.encode(torch.zeros(0, fx["D"]).cuda(), 2)
    assert tuple(e.shape) == (0, fx["N"]) and e.dtype == torch.uint8
    assert tuple(q.decode(e).shape) == (0, fx["D"])
    lead = q.encode(torch.from_numpy(x[:24]).cuda().reshape(2, 3, 4, fx["D"]), 1)
    assert tuple(lead.shape) == (2, 3, 4, fx["N"])
    assert np.array_equal(lead.reshape(24, -1).cpu().numpy(), o.encode(x[:24], 1))


def test_exact_ties_and_degenerate_inputs():
    N, K, D = 4, 16, 16
    centers = np.zeros((N, K, D), np.float32)
    centers[:, :, 0] = 1.0
    state = {"centers": centers, "centers_scale": np.float32(0), "logits_scale": np.float32(0),
             "to_logits.weight": np.zeros((N * K, D), np.float32), "to_logits.bias": np.zeros(N * K, np.float32)}
    q = load_quantizer(state, D, K, N)
    o = oracle_of(state)
    x = np.zeros((5, D), np.float32)
    got = q.encode(torch.from_numpy(x).cuda(), 2, as_bytes=False).cpu().numpy()
    assert np.array_equal(got, o.compute_indexes(x, 2)) and (got == 0).all()
    # duplicated codebook entries with real data: ties everywhere between the twins
    sd = gen.synthetic_state(5, 32, 32, 4)
    sd["centers"][:, 16:] = sd["centers"][:, :16]
    sd["to_logits.weight"] = sd["to_logits.weight"].reshape(4, 32, 32)
    sd["to_logits.weight"][:, 16:] = sd["to_logits.weight"][:, :16]
    sd["to_logits.weight"] = sd["to_logits.weight"].reshape(128, 32)
    sd["to_logits.bias"].reshape(4, 32)[:, 16:] = sd["to_logits.bias"].reshape(4, 32)[:, :16]
    q = load_quantizer(sd, 32, 32, 4)
    o = oracle_of(sd)
    x = gen.make_gaussian(3, 300, 32)
    got = q.encode(torch.from_numpy(x).cuda(), 3, as_bytes=False).cpu().numpy()
    assert np.array_equal(got, o.compute_indexes(x, 3))
    assert (got < 16).all()
    # the same with 16-entry codebooks (k_tf_stage0_k16: four codebooks per wave, the truncation is a rank within a DPP row)
    for N16 in (4, 8, 16):
        sd = gen.synthetic_state(6, 24, 16, N16)
        sd["centers"][:, 8:] = sd["centers"][:, :8]
        w = sd["to_logits.weight"].reshape(N16, 16, 24)
        w[:, 8:] = w[:, :8]
        sd["to_logits.weight"] = w.reshape(N16 * 16, 24)
        sd["to_logits.bias"].reshape(N16, 16)[:, 8:] = sd["to_logits.bias"].reshape(N16, 16)[:, :8]
        q = load_quantizer(sd, 24, 16, N16)
        o = oracle_of(sd)
        x = gen.make_gaussian(4, 333, 24)
        got = q.encode(torch.from_numpy(x).cuda(), 3, as_bytes=False).cpu().numpy()
        assert np.array_equal(got, o.compute_indexes(x, 3))
        assert (got < 8).all()


def test_full_size_properties_config_b():
    """BASELINE config B (dim 512, 8 bytes, B = 65,536): size-independent properties."""
    D, K, N, B = 512, 256, 8, 65536
    sd = gen.synthetic_state(103, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(1234, B, D)
    xd = torch.from_numpy(x).cuda()
    codes = q.encode(xd, 5)
    assert tuple(codes.shape) == (B, N) and codes.dtype == torch.uint8
    # (1) a random sample of rows against the oracle, bit-exact
    rows = np.random.RandomState(0).choice(B, 768, replace=False)
    assert np.array_equal(codes.cpu().numpy()[rows], o.encode(x[rows], 5))
    # (2) the result does not depend on how the batch is cut (chunk independence)
    part = torch.cat([q.encode(xd[:1000], 5), q.encode(xd[1000:4099], 5)])
    assert torch.equal(part, codes[:4099])
    # (3) encode(decode(c)) reproduces... decode is exact, and refinement does not increase the error much:
    y5 = q.decode(codes)
    y0 = q.decode(q.encode(xd, 0))
    e5 = ((y5 - xd) ** 2).sum(dim=1)
    e0 = ((y0 - xd) ** 2).sum(dim=1)
    assert float(e5.sum()) < float(e0.sum()), (float(e5.sum()), float(e0.sum()))
    # (4) decode is linear in the one-hot selection: sum of single-codebook decodes
    rel = float(e5.sum() / (xd ** 2).sum())
    assert 0.0 < rel < 2.0   # untrained synthetic codebooks: sanity only


def test_full_size_config_d():
    """BASELINE config D at its full size (dim 1024, 16 bytes, B = 65,536): sampled rows vs the oracle, chunk
    independence, decode bit-exact, and the reference fixture's 512 rows (same seeded state) inside the big batch."""
    D, K, N, B = 1024, 256, 16, 65536
    fx = fixtures.load("config_d_d1024_n16")
    sd = fx["state"]
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(4321, B, D)
    x[1000:1000 + fx["B"]] = fx["x"]                      # the fixture's rows ride inside the full batch
    xd = torch.from_numpy(x).cuda()
    codes = q.encode(xd, 5)
    assert tuple(codes.shape) == (B, N) and codes.dtype == torch.uint8
    c = codes.cpu().numpy()
    rows = np.random.RandomState(0).choice(B, 512, replace=False)
    assert np.array_equal(c[rows], o.encode(x[rows], 5))
    fixtures.check_codes(fx, 5, c[1000:1000 + fx["B"]], "config D full size (HIP vs reference fixture)")
    part = torch.cat([q.encode(xd[:777], 5), q.encode(xd[777:3000], 5)])
    assert torch.equal(part, codes[:3000])
    y = q.decode(codes)
    assert np.array_equal(y[rows].cpu().numpy(), o.decode(c[rows]))
    e5 = float(((y - xd) ** 2).sum())
    e0 = float(((q.decode(q.encode(xd, 0)) - xd) ** 2).sum())
    assert e5 < e0


def test_config_c_shard_of_one_million_vectors():
    """BASELINE config C's per-GPU share (dim 512, 8 bytes, 1,048,576 vectors = 16 chunks of 65,536 through
    mcq_encode's chunk loop): rows of the first / a middle / the last chunk vs the oracle, equality with
    per-chunk calls, and the reference fixture's rows placed across a chunk boundary."""
    D, K, N, B = 512, 256, 8, 1 << 20
    fx = fixtures.load("config_b_d512_n8")
    sd = fx["state"]
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    xd = torch.randn(B, D, generator=g, device="cuda")
    lo = 65536 * 3 - 2000                                     # 4,096 fixture rows straddle the chunk 2 / chunk 3 border
    xd[lo:lo + fx["B"]] = torch.from_numpy(fx["x"]).cuda()
    codes = q.encode(xd, 5)
    assert tuple(codes.shape) == (B, N)
    fixtures.check_codes(fx, 5, codes[lo:lo + fx["B"]].cpu().numpy(), "1M shard (HIP vs reference fixture)")
    rs = np.random.RandomState(5)
    for c0 in (0, 7, 15):
        rows = c0 * 65536 + rs.choice(65536, 96, replace=False)
        want = o.encode(xd[rows].cpu().numpy(), 5)
        assert np.array_equal(codes[rows].cpu().numpy(), want), c0
    for c0 in (0, 9, 15):                                     # the chunk loop equals separate calls
        sl = slice(c0 * 65536, (c0 + 1) * 65536)
        assert torch.equal(q.encode(xd[sl], 5), codes[sl]), c0
    y = q.decode(codes)                                       # LDS-resident decode at this size
    rows = rs.choice(B, 256, replace=False)
    assert np.array_equal(y[rows].cpu().numpy(), o.decode(codes[rows].cpu().numpy()))


def test_cpu_tensor_is_rejected_loudly():
    fx = fixtures.load("synth_d32_k256_n2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    with pytest.raises(Exception):
        q.encode(torch.from_numpy(fx["x"][:4]), 1)


def test_refine_indexes_from_given_start():
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    o = oracle_of(fx["state"])
    x = fx["x"][:64]
    rs = np.random.RandomState(3)
    start = rs.randint(0, fx["K"], size=(64, fx["N"])).astype(np.int64)     # arbitrary starting indexes
    got = q._refine_indexes(torch.from_numpy(x).cuda(), torch.from_numpy(start).cuda())
    assert got.dtype == torch.int64
    want = np.stack([o.refine_trace(x[i], start[i])["idx"] for i in range(64)])
    assert np.array_equal(got.cpu().numpy(), want)
    # iterating it from the argmax start reproduces _compute_indexes
    idx = q._compute_indexes(torch.from_numpy(x).cuda(), 0)
    for _ in range(2):
        idx = q._refine_indexes(torch.from_numpy(x).cuda(), idx)
    assert torch.equal(idx, q._compute_indexes(torch.from_numpy(x).cuda(), 2))


@pytest.mark.parametrize("D,K,N", [(1, 16, 2), (3, 32, 4), (17, 256, 2), (100, 16, 8), (130, 256, 8), (1000, 64, 8), (778, 32, 16)])
def test_odd_dims_vs_oracle(D, K, N):
    sd = gen.synthetic_state(900 + D, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(901 + D, 257, D)
    for it in (0, 1, 3):
        got = q.encode(torch.from_numpy(x).cuda(), it, as_bytes=False).cpu().numpy()
        assert np.array_equal(got, o.compute_indexes(x, it)), (D, K, N, it)
    codes = o.encode(x, 3)
    assert np.array_equal(q.decode(torch.from_numpy(codes).cuda()).cpu().numpy(), o.decode(codes))


def test_small_workspace_forces_chunks_same_codes():
    """mcq_encode cuts the batch into chunks that fit the caller's workspace; codes do not depend on it."""
    import ctypes
    from quantization_amd import _lib
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    L = _lib.lib()
    N, K, D = fx["N"], fx["K"], fx["D"]
    x = torch.from_numpy(fx["x"][:1000]).cuda()
    whole = q.encode(x, 2)
    small = L.mcq_encode_workspace_bytes(130, N, K, D)         # room for ~130 vectors -> chunks of 128
    ws = torch.empty(small, dtype=torch.uint8, device="cuda")
    out = torch.empty((1000, N), dtype=torch.uint8, device="cuda")
    rc = L.mcq_encode(x.data_ptr(), 1000, q._prepared().data_ptr(), q._lscale_exp, N, K, D, 2, out.data_ptr(), None,
                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out, whole)
    # a workspace too small for 64 vectors is refused, not silently misused
    rc = L.mcq_encode(x.data_ptr(), 1000, q._prepared().data_ptr(), q._lscale_exp, N, K, D, 2, out.data_ptr(), None,
                      ws.data_ptr(), 4096 + 10, torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.MCQ_EWORKSPACE


@pytest.mark.parametrize("name", ["trained_d64_b8_p2", "trained_d64_b4_p1", "trained_d64_b8_p1", "synth_d32_k256_n1", "synth_d64_k256_n16",
                                  "synth_d32_k16_n64", "config_a_d256_n4"])
def test_fixed_point_skipping_gives_identical_codes(name):
    fx = fixtures.load(name)
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    x = torch.from_numpy(fx["x"]).cuda()
    for it in (1, 2, 5, 8):
        q.skip_fixed_points = False
        ref = q.encode(x, it, as_bytes=False)
        q.skip_fixed_points = True
        got = q.encode(x, it, as_bytes=False)
        assert torch.equal(ref, got), (name, it)
    q.skip_fixed_points = True
    assert torch.equal(q.encode(x[:77], 5), q.encode(x, 5)[:77])


@pytest.mark.parametrize("D,N,B", [(37, 8, 1), (512, 8, 4097), (200, 8, 530), (37, 16, 1), (512, 16, 4097), (300, 16, 9)])
def test_passes_from_the_lds_resident_gram_matrix_vs_oracle(D, N, B):
    """k_tf_pass16<N>: 8 or 16 codebooks of 16 entries, every pass of the call in one launch (persistent workgroups, grid-stride
    over the vectors: B = 1, a ragged tail, more vectors than one round of waves); the packed bytes come from the same launch"""
    sd = gen.synthetic_state(900 + D + N, D, 16, N)
    q = load_quantizer(sd, D, 16, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(11 + D + B, B, D)
    x[B // 2] = 0
    xg = torch.from_numpy(x).cuda()
    for it in (1, 3, 6):
        got = q.encode(xg, it, as_bytes=False).cpu().numpy()
        assert np.array_equal(got, o.compute_indexes(x, it)), (D, N, B, it)
    assert np.array_equal(q.encode(xg, 2, as_bytes=True).cpu().numpy(), o.encode(x, 2, as_bytes=True))


@pytest.mark.parametrize("D,K,N,B,it", [(1024, 16, 64, 48, 1), (768, 256, 32, 48, 1), (2048, 256, 8, 100, 2),
                                         (1000, 128, 16, 64, 2)])
def test_large_and_unusual_shapes_vs_oracle(D, K, N, B, it):
    """windowed old-row staging, streaming heavy-L pair path, N = 32 / 64 ladders, dims past 1024"""
    sd = gen.synthetic_state(500 + D + N, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(7 + D, B, D)
    got = q.encode(torch.from_numpy(x).cuda(), it, as_bytes=False).cpu().numpy()
    want = o.compute_indexes(x, it)
    assert np.array_equal(got, want)
    assert np.array_equal(q.decode(torch.from_numpy(got).cuda()).cpu().numpy(), o.decode(want))


@pytest.mark.parametrize("D,K,N,B", [(64, 256, 32, 20000), (48, 16, 64, 30000), (96, 64, 32, 3000), (40, 32, 16, 9000)])
def test_many_codebooks_in_chunks(D, K, N, B):
    """N >= 32 goes through the generic group-table levels (k_tf_up / k_tf_comb, streaming arg-min for 64 x 64) and its
    per-vector workspace is large, so the default chunk is below 65,536 vectors: sampled rows vs the oracle, equality
    with separate calls, decode bit-exact."""
    sd = gen.synthetic_state(800 + N + K, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(33 + D, B, D)
    xd = torch.from_numpy(x).cuda()
    got = q.encode(xd, 2, as_bytes=False)
    rows = np.random.RandomState(1).choice(B, 160, replace=False)
    want = o.compute_indexes(x[rows], 2)
    assert np.array_equal(got.cpu().numpy()[rows], want)
    assert torch.equal(got[:1500], q.encode(xd[:1500], 2, as_bytes=False))
    assert torch.equal(got[B - 777:], q.encode(xd[B - 777:], 2, as_bytes=False))
    assert np.array_equal(q.decode(got[rows]).cpu().numpy(), o.decode(want.astype(np.uint8)))


def test_non_finite_inputs_terminate_with_codes_in_range():
    """NaN / Inf rows give unspecified codes but never a hang, a fault or an
    out-of-range index, and do not disturb their neighbours."""
    sd = gen.synthetic_state(1, 64, 256, 8)
    q = load_quantizer(sd, 64, 256, 8)
    o = oracle_of(sd)
    x = gen.make_gaussian(2, 300, 64)
    bad = x.copy()
    bad[3] = np.nan
    bad[7, 5] = np.inf
    bad[9] = -np.inf
    bad[11] = 1e30
    for skip in (False, True):
        q.skip_fixed_points = skip
        c = q.encode(torch.from_numpy(bad).cuda(), 5, as_bytes=False)
        y = q.decode(c)
        torch.cuda.synchronize()
        c = c.cpu().numpy()
        assert c.min() >= 0 and c.max() <= 255 and tuple(y.shape) == (300, 64)
        good = np.setdiff1d(np.arange(300), [3, 7, 9, 11])
        assert np.array_equal(c[good], o.compute_indexes(x[good], 5))


def test_encode_from_host_overlapped_copies():
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    x = torch.from_numpy(fx["x"])
    want = q.encode(x.cuda(), 3).cpu()
    got = q.encode_from_host(x, 3, chunk=300)          # 7 chunks, ragged tail, double-buffered H2D
    assert got.dtype == torch.uint8 and not got.is_cuda and torch.equal(got, want)
    got64 = q.encode_from_host(x.reshape(4, 512, -1), 3, as_bytes=False, chunk=1000)
    assert tuple(got64.shape) == (4, 512, fx["N"]) and torch.equal(got64.reshape(-1, fx["N"]), want.to(torch.int64))


def test_encode_and_decode_are_hip_graph_capturable():
    """every entry point only enqueues on the given stream (include/mcq.h): a captured encode+decode
    replays on new input without host work, device-side scale factors included"""
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    x_all = torch.from_numpy(fx["x"]).cuda()
    static_x = x_all[:1024].clone()
    want0 = q.encode(static_x, 5)                       # also warms the prepared state and the workspace
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        q.encode(static_x, 5)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_codes = q.encode(static_x, 5)
        static_y = q.decode(static_codes)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_codes, want0)
    static_x.copy_(x_all[1024:2048])
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_codes, q.encode(x_all[1024:2048], 5))
    assert torch.equal(static_y, q.decode(static_codes))


@pytest.mark.parametrize("name", ["trained_d64_b8_p2", "trained_d64_b4_p1", "synth_d30_k32_n4", "config_b_d512_n8"])
def test_fp16_frames_are_widened_in_the_load_path(name):
    """MCQ_ENCODE_X_FP16: fp16 input gives exactly the codes of the same values widened to fp32 (oracle-checked),
    on aligned and unaligned (D % 4 != 0) rows, with and without fixed-point skipping, from host memory too"""
    fx = fixtures.load(name)
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    xh = torch.from_numpy(fx["x"][:1000]).to(torch.float16)
    oq = OracleQuantizer.from_state_dict(fx["state"])
    want = oq.compute_indexes(xh.float().numpy(), 5)
    got = q.encode(xh.cuda(), 5, as_bytes=False)
    assert np.array_equal(got.cpu().numpy(), want)
    assert torch.equal(got, q.encode(xh.cuda().float(), 5, as_bytes=False))
    assert torch.equal(q.encode(xh.cuda()[1:], 5, as_bytes=False), got[1:])      # rows at an odd 2-byte offset
    q.skip_fixed_points = True
    assert torch.equal(q.encode(xh.cuda(), 5, as_bytes=False), got)
    q.skip_fixed_points = False
    assert torch.equal(q.encode_from_host(xh, 5, as_bytes=False, chunk=300).cuda(), got)


@pytest.mark.parametrize("D,K,N", [(40, 64, 8), (30, 32, 4), (100, 256, 2), (512, 256, 8), (1024, 256, 16), (260, 64, 32), (64, 16, 8),
                                   (1000, 64, 8), (100, 128, 16), (36, 256, 8), (72, 256, 16)])
def test_decode_of_big_batches_takes_the_xcd_sliced_kernel_and_stays_bit_exact(D, K, N):
    """B >= 4096 unpacked codes go through k_decode_sliced (feature axis cut in eight, one slice per XCD):
    same sums in the same order as the oracle, for ragged dims, uint8 and int64 codes, ragged batch sizes"""
    state = gen.synthetic_state(31, D, K, N)
    q = load_quantizer(state, D, K, N)
    oq = OracleQuantizer.from_state_dict(state)
    B = 5003
    rng = np.random.default_rng(5)
    codes = rng.integers(0, K, size=(B, N), dtype=np.int64)
    want = oq.decode(codes.astype(np.uint8))
    got8 = q.decode(torch.from_numpy(codes.astype(np.uint8)).cuda())
    got64 = q.decode(torch.from_numpy(codes).cuda())
    assert np.array_equal(got8.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert torch.equal(got8, got64)
    # the LDS-resident kernels of large batches (threshold lowered through its test hook): the block-staged one for packed byte
    # codes of 4 / 8 / 16 codebooks (here only its partial-block path: 5,003 vectors), k_decode_lds for the rest; dims whose last
    # slice reaches into the padding must not write past a row
    os.environ["MCQ_DECODE_LDS_MIN"] = "4096"
    try:
        lds8 = q.decode(torch.from_numpy(codes.astype(np.uint8)).cuda())
        lds64 = q.decode(torch.from_numpy(codes).cuda())
    finally:
        del os.environ["MCQ_DECODE_LDS_MIN"]
    assert torch.equal(lds8, got8) and torch.equal(lds64, got8)


@pytest.mark.parametrize("D", [72, 256])
def test_decode_of_sixteen_big_codebooks_uses_32_byte_slices(D):
    """16 x 256 codebooks: the 64-byte slices of all rows (256 KB) do not fit the LDS; k_decode_blk keeps 32-byte slices there
    instead (whole code blocks and a partial one per workgroup at this size) -- same sums, n ascending"""
    state = gen.synthetic_state(33, D, 256, 16)
    q = load_quantizer(state, D, 256, 16)
    oq = OracleQuantizer.from_state_dict(state)
    B = 33001
    codes = np.random.default_rng(6).integers(0, 256, size=(B, 16), dtype=np.uint8)
    got = q.decode(torch.from_numpy(codes).cuda()).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oq.decode(codes).view(np.uint32))


@pytest.mark.parametrize("D,K,N,B", [(40, 64, 8, 70001), (512, 256, 8, 65536), (256, 256, 4, 140003), (36, 256, 8, 33333), (100, 128, 16, 40001),
                                     (512, 32, 4, 16384)])
def test_block_staged_decode_whole_and_partial_blocks(D, K, N, B):
    """k_decode_blk (packed byte codes of 4 / 8 / 16 codebooks, >= 16,384 vectors): the codes of a block reach the LDS by LDS-DMA
    a block ahead and the wait for them is a counted one (the block's stores stay in flight).  Batches that give a workgroup
    several whole blocks and a partial one, ragged dims, every row against the oracle; the other kernels must agree"""
    state = gen.synthetic_state(35, D, K, N)
    q = load_quantizer(state, D, K, N)
    oq = OracleQuantizer.from_state_dict(state)
    codes = np.random.default_rng(8).integers(0, K, size=(B, N), dtype=np.uint8)
    cd = torch.from_numpy(codes).cuda()
    got = q.decode(cd)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), oq.decode(codes).view(np.uint32))
    for _ in range(3):                                   # back to back into fresh buffers: no stale block of codes
        assert torch.equal(q.decode(cd), got)
    os.environ["MCQ_DECODE_BLK"] = "0"
    try:
        other = q.decode(cd)
    finally:
        del os.environ["MCQ_DECODE_BLK"]
    assert torch.equal(other, got)
    assert torch.equal(q.decode(cd[1:]), got[1:])        # codes at an odd offset: not 16-byte aligned, another kernel
    assert torch.equal(q.decode(cd.long()), got)         # int64 indexes (what _compute_indexes returns)


def test_derived_state_follows_fused_optimizer_steps():
    """Adam(fused=True) updates parameters without bumping Tensor._version; the cached prepared state must not go stale"""
    fx = fixtures.load("trained_d64_b4_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    x = torch.from_numpy(fx["x"][:512]).cuda()
    before = q.encode(x, 2)
    opt = torch.optim.Adam(q.parameters(), lr=0.05, fused=True)
    with torch.enable_grad():
        for p in q.parameters():
            p.grad = torch.randn_like(p)
    opt.step()
    after = q.encode(x, 2)
    q2 = load_quantizer({k: v.detach().cpu().numpy() for k, v in q.state_dict().items()}, fx["D"], fx["K"], fx["N"])
    assert torch.equal(after, q2.encode(x, 2)) and not torch.equal(after, before)
    # an optimizer that holds no quantizer parameter leaves the cached state alone (frozen quantizer inside
    # another model's training loop)
    cached = q._prepared()
    other = torch.nn.Linear(4, 4).cuda()
    opt2 = torch.optim.Adam(other.parameters(), fused=True)
    for p in other.parameters():
        p.grad = torch.ones_like(p)
    opt2.step()
    assert q._prepared() is cached
    with torch.no_grad():
        q.centers.data.mul_(-1.0)          # unversioned edit: the documented escape hatch
    q.invalidate_cache()
    q3 = load_quantizer({k: v.detach().cpu().numpy() for k, v in q.state_dict().items()}, fx["D"], fx["K"], fx["N"])
    assert torch.equal(q.encode(x, 2), q3.encode(x, 2))


def test_encode_does_not_depend_on_the_callers_autograd_mode():
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    x = torch.from_numpy(fx["x"][:777]).cuda()
    want = q.encode(x, 3)
    with torch.enable_grad():
        assert all(p.requires_grad for p in q.parameters())
        got = q.encode(x, 3)
        got_idx = q._compute_indexes(x, 3)
    assert torch.equal(got, want) and torch.equal(got_idx.to(torch.uint8), want)
    fixtures.check_codes(fx, 5, q.encode(torch.from_numpy(fx["x"]).cuda(), 5).cpu().numpy(), "enable_grad encode")


def test_logits_refine_is_logits_argmax_then_refine_indexes():
    """mcq_logits_refine (one call: the frames become limb planes once, the indexes stay bytes) gives the logits of
    mcq_logits_argmax and the indexes mcq_refine_indexes returns from that arg max"""
    from quantization_amd import _lib
    fx = fixtures.load("trained_d64_b8_p2")
    q = load_quantizer(fx["state"], fx["D"], fx["K"], fx["N"])
    N, K, D = fx["N"], fx["K"], fx["D"]
    L = _lib.lib()
    x = torch.from_numpy(fx["x"][:777]).cuda()
    B = x.shape[0]
    with torch.no_grad():
        blob = q._prepared()
    ws = q._workspace(B, x.device)
    st = torch.cuda.current_stream().cuda_stream
    lg1 = torch.empty((B, N * K), device="cuda")
    lg2 = torch.empty_like(lg1)
    a1 = torch.empty((B, N), dtype=torch.int64, device="cuda")
    i1 = torch.empty_like(a1)
    i2 = torch.empty_like(a1)
    for iters in (0, 1, 3):
        assert L.mcq_logits_argmax(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, lg1.data_ptr(), a1.data_ptr(),
                                   ws.data_ptr(), ws.numel(), st, 0) == 0
        assert L.mcq_refine_indexes(x.data_ptr(), B, blob.data_ptr(), N, K, D, iters, a1.data_ptr(), i1.data_ptr(),
                                    ws.data_ptr(), ws.numel(), st) == 0
        assert L.mcq_logits_refine(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, iters, lg2.data_ptr(),
                                   i2.data_ptr(), ws.data_ptr(), ws.numel(), st, 0) == 0
        torch.cuda.synchronize()
        assert torch.equal(lg1, lg2) and torch.equal(i1, i2), iters
        assert torch.equal(i2, q._compute_indexes(x, iters))


@pytest.mark.parametrize("per_lane,cnt", [(1, 8), (4, 8), (4, 16), (4, 32), (16, 16), (16, 32), (16, 64), (4, 1), (16, 1),
                                          (-4, 16), (-4, 32), (-16, 32), (-4, 64), (-1004, 16), (-1004, 64)])
def test_wave_selection_paths(per_lane, cnt):
    """wave_select_set on its own (mcq_test_select): the cnt smallest of 64 * per_lane scores by (value, position), listed in
    ascending position (oracle/mcq_oracle.c::select_smallest), for random scores, heavy ties, survivors clustered in a few lanes
    (more than one per lane: the general quickselect takes over) and constant input"""
    from quantization_amd import _lib
    L = _lib.lib()
    # (negative per_lane: the slot-major layout, key i of a lane at position 64 i + lane; -1004: four keys per lane handled as
    # positions in no particular order -- the general form and a rank by position)
    kpl = abs(per_lane) % 1000
    M = 64 * kpl
    rs = np.random.RandomState(abs(per_lane) * 100 + cnt)
    cases = []
    for c in range(40):
        kind = c % 5
        if kind == 0:
            sc = rs.standard_normal(M) * 10 + 500
        elif kind == 1:                                   # heavy ties: a handful of distinct values
            sc = rs.randint(0, 6, size=M).astype(np.float64)
        elif kind == 2:                                   # the small scores all sit in a few lanes
            sc = rs.standard_normal(M) + 100
            lanes = rs.choice(64, size=min(64, max(cnt, 12)), replace=False)
            for l in lanes:
                if per_lane in (-4, -16):
                    sc[l::64] = rs.standard_normal(kpl)
                else:
                    sc[kpl * l:kpl * (l + 1)] = rs.standard_normal(kpl)
        elif kind == 3:
            sc = np.full(M, 3.25)
        else:                                             # negative and positive, zeros
            sc = rs.standard_normal(M)
            sc[rs.randint(0, M, size=M // 8)] = 0.0
        cases.append(sc.astype(np.float32))
    sc = np.stack(cases)
    dsc = torch.from_numpy(sc).cuda()
    ov = torch.zeros((len(cases), 64), dtype=torch.float32, device="cuda")
    op = torch.zeros((len(cases), 64), dtype=torch.int32, device="cuda")
    assert L.mcq_test_select(dsc.data_ptr(), len(cases), per_lane, cnt, ov.data_ptr(), op.data_ptr(), None) == 0
    torch.cuda.synchronize()
    for c in range(len(cases)):
        order = np.sort(np.lexsort((np.arange(M), sc[c]))[:cnt])          # the cnt smallest by (value, position), listed by position
        assert np.array_equal(op[c, :cnt].cpu().numpy(), order), (c, per_lane, cnt)
        assert np.array_equal(ov[c, :cnt].cpu().numpy(), sc[c][order])


def test_decode_only_state_is_small_and_a_later_search_rebuilds():
    """decode() on a quantizer that was never searched with builds the scaled centers only (mcq_prepared_decode_bytes: no
    limb planes, no Gram matrix); a search afterwards replaces it by the full state; decode gives the same bits either way."""
    from quantization_amd import Quantizer, _lib
    L = _lib.lib()
    D, K, N = 96, 256, 8
    torch.manual_seed(2)
    q = Quantizer(D, K, N).cuda()
    codes = torch.randint(0, K, (777, N), device="cuda", dtype=torch.uint8)
    with torch.no_grad():
        y0 = q.decode(codes)
        assert q._prep.flavour == "decode" and q._prep.blob.numel() == L.mcq_prepared_decode_bytes(N, K, D)
        assert L.mcq_prepared_decode_bytes(N, K, D) < L.mcq_prepared_bytes(N, K, D) // 8
        x = torch.randn(300, D, device="cuda")
        c = q.encode(x, 2)
        assert q._prep.flavour == "host" and q._prep.blob.numel() == L.mcq_prepared_bytes(N, K, D)
        y1 = q.decode(codes)
        assert q._prep.flavour == "host"
    assert torch.equal(y0, y1)
    ref = q.get_centers().detach()[torch.arange(N, device="cuda"), codes.long()]      # (B, N, D), summed in codebook order
    acc = ref[:, 0]
    for n in range(1, N):
        acc = acc + ref[:, n]
    assert torch.equal(y0, acc) and tuple(c.shape) == (300, N)


def test_config_e_phase_one_state_bit_exact_vs_oracle():
    """The quantizer the REFERENCE's trainer starts BASELINE config E with (16 codebooks of 16 entries at dim 512: the initial
    state of tests/golden/trainer_config_e_d512_b8.npz) on a config-E batch of 4,096 frames: codes for 0, 1 and 2 passes -- what a
    training step asks for -- bit-exact against the oracle, as int64 and as packed bytes."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_config_e_d512_b8.npz"))
    state = {k[len("init."):]: fx[k] for k in fx.files if k.startswith("init.") and k != "init.id_buf"}
    D, K, N = 512, 16, 16
    assert state["centers"].shape == (N, K, D)
    q = load_quantizer(state, D, K, N)
    o = oracle_of(state)
    x = gen.make_x(int(fx["data_seed"]), 4096, D)
    xg = torch.from_numpy(x).cuda()
    for it in (0, 1, 2):
        want = o.compute_indexes(x, it)
        got = q.encode(xg, it, as_bytes=False).cpu().numpy()
        assert np.array_equal(got, want), (it, int((got != want).any(axis=1).sum()))
    assert np.array_equal(q.encode(xg, 2).cpu().numpy(), o.encode(x, 2))


def test_full_size_config_b_with_a_reference_trained_state_and_shifted_frames():
    """BASELINE config B's shape with the state the REFERENCE trained at dim 512 / 8 bytes (trained_d512_b8_p2) on 65,536 frames --
    half of them the training distribution, half with a common offset of 10, where the centered tables matter:
    sampled rows bit-exact against the oracle; the fixture's 2,048 rows, riding inside the batch across the middle, against the
    reference's own codes; chunk independence."""
    fx = fixtures.load("trained_d512_b8_p2")
    D, K, N, B = 512, 256, 8, 65536
    q = load_quantizer(fx["state"], D, K, N)
    o = oracle_of(fx["state"])
    x = np.concatenate([gen.make_kind("make_x", 777, B // 2, D), gen.make_kind("mean10", 778, B // 2, D)])
    x[B // 2 - 1024:B // 2 - 1024 + fx["B"]] = fx["x"]
    xd = torch.from_numpy(x).cuda()
    codes = q.encode(xd, 5)
    c = codes.cpu().numpy()
    rows = np.random.RandomState(3).choice(B, 640, replace=False)
    assert np.array_equal(c[rows], o.encode(x[rows], 5))
    fixtures.check_codes(fx, 5, c[B // 2 - 1024:B // 2 - 1024 + fx["B"]], "trained d512 state inside a full batch (HIP vs reference fixture)")
    part = torch.cat([q.encode(xd[B // 2 - 500:B // 2], 5), q.encode(xd[B // 2:B // 2 + 700], 5)])
    assert torch.equal(part, codes[B // 2 - 500:B // 2 + 700])
    # the shifted half is far from what the quantizer was trained on: refinement must still not lose against the initial guess
    y5, y0 = q.decode(codes), q.decode(q.encode(xd, 0))
    assert float(((y5 - xd) ** 2).sum()) < float(((y0 - xd) ** 2).sum())


# ---------------------------------------------------------------- codebooks of 512 / 1,024 entries
# Quantizer(codebook_size=512 / 1024) is outside what QuantizerTrainer produces but inside what the reference's Quantizer accepts
# (quantization.py:35 asks for a power of two; only the byte form needs <= 256, :271).  Entries are held in two bytes on that path
# (mcq_tf_kernels.h, CT); the fixtures k512_* / k1024_* above pin it to the reference, these cases to the oracle at more shapes.
@pytest.mark.parametrize("D,K,N,B,it", [(100, 512, 8, 300, 3), (64, 1024, 4, 257, 2), (16, 512, 32, 64, 1), (33, 1024, 16, 96, 2),
                                         (20, 1024, 1, 130, 1), (512, 512, 2, 64, 5)])
def test_codebooks_of_512_and_1024_entries_vs_oracle(D, K, N, B, it):
    sd = gen.synthetic_state(1200 + D + N, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(7201 + D, B, D)
    xd = torch.from_numpy(x).cuda()
    got = q.encode(xd, it, as_bytes=False)
    assert got.dtype == torch.int64
    want = o.compute_indexes(x, it)
    assert want.dtype == np.uint16 and int(want.max()) > 255
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(q.logits_kernel(xd).cpu().numpy(), o.logits(x))
    assert np.array_equal(q.decode(got).cpu().numpy(), o.decode(want))
    with pytest.raises(AssertionError):
        q.encode(xd, it)                                    # as_bytes=True: quantization.py:271
    # the search from caller-supplied indexes, and fixed-point skipping
    start = np.random.RandomState(5).randint(0, K, size=(B, N)).astype(np.int64)
    r1 = q._refine_indexes(xd, torch.from_numpy(start).cuda()).cpu().numpy()
    assert np.array_equal(r1[:8], np.stack([o.refine_trace(x[i], start[i])["idx"] for i in range(8)]))
    q.skip_fixed_points = True
    assert torch.equal(q.encode(xd, max(it, 2), as_bytes=False), torch.from_numpy(o.compute_indexes(x, max(it, 2)).astype(np.int64)).cuda())


def test_wide_codebooks_in_chunks_and_the_product_of_two_32_entry_codebooks():
    D, K, N, B = 64, 512, 8, 70000                          # past the default chunk of 65,536 vectors
    sd = gen.synthetic_state(1300, D, K, N)
    q = load_quantizer(sd, D, K, N)
    o = oracle_of(sd)
    x = gen.make_gaussian(1301, B, D)
    xd = torch.from_numpy(x).cuda()
    got = q.encode(xd, 2, as_bytes=False)
    rows = np.concatenate([np.arange(65536 - 40, 65536 + 40), np.random.RandomState(2).choice(B, 80, replace=False)])
    want = o.compute_indexes(x[rows], 2)
    assert np.array_equal(got.cpu().numpy()[rows], want)
    assert torch.equal(got[B - 999:], q.encode(xd[B - 999:], 2, as_bytes=False))
    y = q.decode(got)                                        # 70,000 rows of int64 indexes
    assert np.array_equal(y.cpu().numpy()[rows], o.decode(want))
    # get_product_quantizer (:81-112) of 4 x 32 gives 2 x 1,024: same reconstruction for paired indexes, and it encodes
    sd2 = gen.synthetic_state(1302, 40, 32, 4)
    q2 = load_quantizer(sd2, 40, 32, 4)
    p = q2.get_product_quantizer()
    assert (p.codebook_size, p.num_codebooks) == (1024, 2)
    x2 = gen.make_gaussian(1303, 200, 40)
    i2 = q2.encode(torch.from_numpy(x2).cuda(), 2, as_bytes=False)
    paired = torch.stack([i2[:, 0] * 32 + i2[:, 1], i2[:, 2] * 32 + i2[:, 3]], dim=1)
    assert torch.allclose(p.decode(paired), q2.decode(i2), rtol=0, atol=1e-5)
    po = OracleQuantizer(p.centers.detach().cpu().numpy(), float(p.centers_scale), p.to_logits.weight.detach().cpu().numpy(),
                         p.to_logits.bias.detach().cpu().numpy(), float(p.logits_scale))
    assert np.array_equal(p.encode(torch.from_numpy(x2).cuda(), 3, as_bytes=False).cpu().numpy(), po.compute_indexes(x2, 3))


def test_compute_loss_of_a_512_entry_quantizer_follows_the_reference():
    """compute_loss (:184-242) outside the trainer's shapes: 4 x 512.  The fused loss kernels stop at 256 entries; the module then
    forms the same sums from its HIP index search / decode and torch ops.  Losses and gradients against the reference's own
    (tests/golden/make_golden_loss_wide.py)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_k512.npz"))
    D, K, N, B = (int(z[k]) for k in "DKNB")
    sd = gen.synthetic_state(int(z["state_seed"]), D, K, N)
    q = load_quantizer(sd, D, K, N)
    x = torch.from_numpy(gen.make_x(int(z["x_seed"]), B, D)).cuda()
    for iters in (0, 2):
        q.zero_grad()
        with torch.enable_grad():           # (this module's tests run under no_grad)
            losses = q.compute_loss(x, iters)
            (losses[0] + 0.3 * losses[1] + 0.2 * losses[2] + 0.1 * losses[3]).backward()
        got = np.array([float(v) for v in losses])
        assert np.allclose(got, z[f"losses_it{iters}"], rtol=1e-4, atol=1e-6), (iters, got, z[f"losses_it{iters}"])
        for name, p in q.named_parameters():
            want = z[f"grad_it{iters}.{name}"]
            g = p.grad.detach().cpu().numpy()
            assert np.linalg.norm(g - want) <= 1e-3 * np.linalg.norm(want) + 1e-7, (iters, name)
