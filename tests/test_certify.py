"""Every row on which the oracle's codes differ from the reference's, certified (tests/golden/certify.py): the pass and the node of the
combine tree where the two part, the fp64 gap at that decision relative to the two competing scores (< 1e-6), and the
reconstruction errors of the two results (within 2 %).  The rows are FOUND here, not listed: the oracle runs over every fixture."""
import numpy as np
import pytest

from golden import certify, fixtures
from oracle.oracle import OracleQuantizer

# (the fixtures on which round 5's review found the mismatches, plus the one with the tightest margins; test_oracle_golden.py runs
# check_codes -- and with it the certification of whatever differs -- over ALL fixtures)
NAMES = ["stress_mean10_d512_b8_p2", "stress_outlier300_d64_b4_p2", "stress_mean10_d64_b8_p1", "synth_d16_k256_n64", "trained_d512_b8_p2"]


@pytest.mark.parametrize("name", NAMES)
def test_mismatching_rows_are_certified_near_ties(name, capsys):
    fx = fixtures.load(name)
    s = fx["state"]
    o = OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"], float(s["logits_scale"]), scales_exp=getattr(s, "scales_exp", None))
    found = 0
    for it in fx["iters"]:
        codes = np.asarray(o.compute_indexes(fx["x"], it)).reshape(fx[f"codes_it{it}"].shape)
        for row in np.flatnonzero((codes != fx[f"codes_it{it}"]).any(axis=1)):
            c = certify.certify_row(fx, it, int(row), codes[row])
            found += 1
            with capsys.disabled():
                print(f"\n  {name} iters={it} row {row}: {c['stage']}; gap {c['gap']:.2e} (< {certify.NEAR:g}); "
                      f"SSE of the oracle's code {c['sse_rel']:+.2e} of the reference's")
            assert c["gap"] < certify.NEAR and abs(c["sse_rel"]) <= certify.OUTCOME
    assert found >= 1, "this fixture was listed because the oracle differs from the reference on it"


def test_margins_bite_on_offset_fixtures():
    """The near-tie flag must leave most rows of an offset fixture OUTSIDE (the round-1 normalisation flagged all 2,048)."""
    for name, cap in (("stress_mean10_d512_b8_p2", 0.5), ("stress_mean100_d64_b4_p2", 0.1), ("stress_mean10_d64_b8_p2", 0.2)):
        fx = fixtures.load(name)
        it = fx["iters"][-1]
        share = float((fx[f"margin2_it{it}"] < fixtures.NEAR_TIE).mean())
        old = float((fx[f"margin_it{it}"] < fixtures.NEAR_TIE).mean())
        assert share < cap, (name, share)
        assert old > 0.9, (name, old)


def test_a_clear_difference_is_refused():
    """certify_row must fail on a code that is NOT a near-tie outcome: the reference's row with one entry replaced by a poor one"""
    fx = fixtures.load("trained_d64_b8_p2")
    it = fx["iters"][-1]
    row = int(np.argmax(fx[f"margin2_it{it}"]))          # the clearest row of the fixture
    wrong = fx[f"codes_it{it}"][row].astype(np.int64).copy()
    wrong[0] = (wrong[0] + 97) % fx["K"]
    with pytest.raises(AssertionError):
        certify.certify_row(fx, it, row, wrong)
