"""The oracle's fixed-point inner product ("fixdot", oracle/mcq_oracle.c) against exact arithmetic (not gpu).

The reference forms its logits with an fp32 GEMM whose summation order is unspecified; fixdot replaces the order by exact
integer sums of 8-bit limb products.  These tests pin what DESIGN.md section 2 claims for it: an error bounded by the row
maxima (2^-28 max|x| max|w| per term plus the final fp32 rounding), no worse than an fp32 chain on ordinary data, and an
independent restatement of the definition in Python integers."""
import numpy as np
import pytest

from oracle.oracle import OracleQuantizer


def _state(rs, N, K, D, wide=False):
    centers = rs.standard_normal((N, K, D)).astype(np.float32)
    W = (rs.standard_normal((N * K, D)) / np.sqrt(D)).astype(np.float32)
    if wide:      # rows with a large dynamic range: small elements lose relative, never absolute, precision
        W *= np.exp(rs.uniform(-8, 8, size=W.shape)).astype(np.float32)
    bias = rs.standard_normal(N * K).astype(np.float32) * 0.1
    return centers, W, bias


def _fix_row(v):
    """limbs [4][D] (python ints) and the row exponent, as the oracle header defines them"""
    m = float(np.max(np.abs(v))) if len(v) else 0.0
    be = (np.float32(m).view(np.uint32) >> 23) & 0xff
    e = max(int(be), 1) - 126
    q = np.rint(np.clip(np.ldexp(v.astype(np.float64), 30 - e), -2.0 ** 30, 2.0 ** 30)).astype(np.int64)
    limbs, r = [None] * 4, q.copy()
    for i in (3, 2, 1):
        l = ((r & 0xff) ^ 0x80) - 0x80          # signed low byte
        limbs[i] = l
        r = (r - l) >> 8
    limbs[0] = r
    return limbs, e


def _fixdot(a, b):
    la, ea = _fix_row(a)
    lb, eb = _fix_row(b)
    T = [0, 0, 0, 0]
    for i in range(4):
        for j in range(4 - i):
            T[i + j] += int(np.sum(la[i] * lb[j]))
    t = np.float32(T[3])
    for s, w in ((2, 256.0), (1, 65536.0), (0, 16777216.0)):
        t = np.float32(np.float64(np.float32(T[s])) * w + np.float64(t))      # one rounding: an fma of fp32 operands
    return np.float32(np.ldexp(np.float64(t), ea + eb - 36))


@pytest.mark.parametrize("D,K,N,wide", [(512, 256, 2, False), (100, 16, 4, False), (37, 32, 2, True), (1000, 64, 1, True)])
def test_fixdot_error_bound(D, K, N, wide):
    rs = np.random.RandomState(D + K)
    centers, W, bias = _state(rs, N, K, D, wide)
    # scale factors exp(0) = 1, no bias, all-zero centers (the data mean the logits' frames are centered by is then 0 and what the
    # centering takes out, fixdot(mean, W[r]), is 0 as well): the logits are the raw products
    o = OracleQuantizer(np.zeros_like(centers), 0.0, W, np.zeros_like(bias), 0.0)
    x = rs.standard_normal((24, D)).astype(np.float32)
    if wide:
        x *= np.exp(rs.uniform(-6, 6, size=x.shape)).astype(np.float32)
    got = o.logits(x).astype(np.float64)
    exact = x.astype(np.float64) @ W.astype(np.float64).T
    xm = np.abs(x).max(axis=1)[:, None].astype(np.float64)
    wm = np.abs(W).max(axis=1)[None, :].astype(np.float64)
    bound = D * xm * wm * 2.0 ** -27 + np.abs(exact) * 2.0 ** -23
    assert (np.abs(got - exact) <= bound).all(), float((np.abs(got - exact) / bound).max())
    if not wide:      # on ordinary data at least as close as an fp32 chain
        chain = np.zeros((x.shape[0], W.shape[0]), np.float32)
        for d in range(D):
            chain = (chain + x[:, d:d + 1] * W[None, :, d]).astype(np.float32)
        assert np.abs(got - exact).mean() <= np.abs(chain - exact).mean()


def test_fixdot_matches_the_written_definition():
    rs = np.random.RandomState(7)
    D, K, N = 45, 16, 2
    centers, W, bias = _state(rs, N, K, D, wide=True)
    W[3] = 0.0                                   # an all-zero row: exponent -125, product 0
    o = OracleQuantizer(np.zeros_like(centers), 0.0, W, np.zeros_like(bias), 0.0)      # (zero mean: logits = raw products)
    x = rs.standard_normal((5, D)).astype(np.float32)
    x[1] *= 1e-30                                # tiny rows scale exactly
    x[2] *= 1e+20
    got = o.logits(x)
    for b in range(x.shape[0]):
        for r in range(N * K):
            want = _fixdot(x[b], W[r])
            assert got[b, r] == want or (got[b, r] == 0 and want == 0), (b, r, got[b, r], want)
    assert (got[:, 3] == 0).all()
