"""The CPU oracle against the fixtures captured from the reference (not gpu)."""
import numpy as np
import pytest

from golden import fixtures
from oracle.oracle import OracleQuantizer, ladder

ALL = fixtures.names()
SMALL = [n for n in ALL if not n.startswith("config_b") and not n.startswith("config_d")]


def _oracle(fx):
    s = fx["state"]
    return OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"],
                           float(s["logits_scale"]), scales_exp=getattr(s, "scales_exp", None))


def test_fixture_inventory():
    # every ladder of SURVEY.md 8a is covered by at least one fixture
    shapes = {(int(fixtures.load(n)["K"]), int(fixtures.load(n)["N"])) for n in SMALL}
    for need in [(256, 1), (256, 2), (256, 4), (256, 8), (256, 16), (256, 32), (16, 8), (16, 16), (16, 32), (16, 64),
                 (512, 1), (1024, 2), (512, 4), (1024, 8), (512, 16)]:
        assert need in shapes, need


def test_ladders():
    assert ladder(4, 256) == (16, [(16, 16), (16, 1)])
    assert ladder(8, 256) == (16, [(16, 16), (16, 32), (32, 1)])
    assert ladder(16, 256) == (16, [(16, 16), (16, 32), (32, 32), (32, 1)])
    assert ladder(16, 16) == (8, [(8, 8), (8, 16), (16, 16), (16, 1)])
    assert ladder(1, 256) == (1, [])


@pytest.mark.parametrize("name", ALL)
def test_codes_match_reference(name):
    fx = fixtures.load(name)
    o = _oracle(fx)
    for it in fx["iters"]:
        codes = o.compute_indexes(fx["x"], it)
        fixtures.check_codes(fx, it, codes, f"{name} iters={it}")


@pytest.mark.parametrize("name", SMALL)
def test_bytes_and_decode(name):
    fx = fixtures.load(name)
    o = _oracle(fx)
    it = fx["iters"][-1]
    ref_codes = fx[f"codes_it{it}"]
    # (codebooks of more than 256 entries have no byte form -- quantization.py:271 asserts -- their indexes are decoded as they are)
    ref_bytes = fx[f"bytes_it{it}"] if fx["K"] <= 256 else ref_codes
    # packing (quantization.py:266-272) applied to the reference's own indexes
    K, idx = fx["K"], ref_codes.astype(np.int64)
    while K * K <= 256:
        idx = idx[:, ::2] + K * idx[:, 1::2]
        K *= K
    assert np.array_equal(idx.astype(ref_bytes.dtype), ref_bytes)
    assert np.array_equal(o.separate_indexes(ref_bytes), ref_codes)
    # decode (quantization.py:131-148) of the reference's codes, within 1e-5 relative
    y = o.decode(ref_bytes)
    head = fx["decode_head"]
    scale = np.abs(head).max()
    assert np.abs(y[:16] - head).max() <= 1e-5 * scale
    assert np.allclose(y.astype(np.float64).sum(axis=1), fx["decode_rowsum"], rtol=0, atol=1e-5 * scale * fx["D"])
    assert np.allclose((y.astype(np.float64) ** 2).sum(axis=1), fx["decode_rowsumsq"], rtol=1e-5)
    if fx["N"] <= 16:
        assert np.array_equal(y[:16], head)  # same summation order as torch's sum(dim=0)


def test_chunking_and_threads_do_not_change_codes():
    fx = fixtures.load("trained_d64_b8_p2")
    o = _oracle(fx)
    x = fx["x"][:300]
    whole = o.compute_indexes(x, 2, nthreads=8)
    one = o.compute_indexes(x, 2, nthreads=1)
    parts = np.concatenate([o.compute_indexes(x[i:i + 7], 2) for i in range(0, 300, 7)])
    assert np.array_equal(whole, one) and np.array_equal(whole, parts)


def test_empty_batch():
    fx = fixtures.load("synth_d32_k256_n2")
    o = _oracle(fx)
    assert o.compute_indexes(np.zeros((0, 32), np.float32), 3).shape == (0, 2)
    assert o.decode(np.zeros((0, 2), np.uint8)).shape == (0, 32)


def test_exact_ties_pick_lowest_position():
    # duplicate codebook entries and a zero input: every comparison is an exact tie
    N, K, D = 4, 16, 16
    centers = np.zeros((N, K, D), np.float32)
    centers[:, :, 0] = 1.0
    o = OracleQuantizer(centers, 0.0, np.zeros((N * K, D), np.float32), np.zeros(N * K, np.float32), 0.0)
    codes = o.compute_indexes(np.zeros((3, D), np.float32), 2)
    assert (codes == 0).all()
