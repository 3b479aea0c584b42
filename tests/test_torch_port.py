"""The torch-CPU restatement used as bench.py's cpu_baseline reproduces the fixtures,
which ties its timing to the real reference's op sequence."""
import numpy as np
import pytest

from golden import fixtures
from oracle.torch_port import TorchPortQuantizer


@pytest.mark.parametrize("name", ["trained_d64_b4_p1", "trained_d64_b8_p2", "synth_d64_k256_n16", "config_a_d256_n4"])
def test_port_reproduces_reference_codes(name):
    fx = fixtures.load(name)
    p = TorchPortQuantizer(fx["state"])
    it = fx["iters"][-1]
    n = min(fx["B"], 512)
    got = p.encode(fx["x"][:n], it).numpy()
    ref = fx[f"codes_it{it}"][:n]
    # same torch ops on the same machine class: identical except, at most, near-ties
    bad = (got != ref).any(axis=1)
    assert (bad & (fx[f"margin_it{it}"][:n] >= fixtures.NEAR_TIE)).sum() == 0
    assert bad.sum() <= 2
