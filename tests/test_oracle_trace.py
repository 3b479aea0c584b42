"""Stage-by-stage check of the oracle against intermediates captured from the reference's
Quantizer._refine_indexes (quantization/quantization.py:308-547; fixtures trace_*.npz written by
tests/golden/make_golden_trace.py): stage-0 scores, every combined score table, every shortlist."""
import numpy as np
import pytest

from golden import fixtures
from oracle.oracle import OracleQuantizer, ladder

RTOL = 1e-5      # fp32 summation-order noise between MKL and the oracle's fixed fmaf chains


def _close(a, b, scale):
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= RTOL * scale


@pytest.mark.parametrize("name", fixtures.trace_names())
def test_stage_intermediates_match_reference(name):
    z = np.load(f"{fixtures.HERE}/{name}.npz")
    src = fixtures.load(str(z["source"]))
    D, K, N = src["D"], src["K"], src["N"]
    oq = OracleQuantizer.from_state_dict(src["state"])
    first, lad = ladder(N, K)
    keeps = [first] + [ko for (_, ko) in lad]
    assert keeps == list(z["keeps"]), "ladder differs from the reference's sort calls"
    nstage = len(keeps)
    reordered = 0
    for b in range(int(z["nvec"])):
        t = oq.refine_trace(src["x"][b], z["idx_in"][b])
        # ---- stage 0: S[n][k] of :418
        ref0 = z["scores0"][b]
        scale = np.abs(ref0).max()
        assert _close(t["S0"], ref0, scale), f"{name}[{b}]: stage-0 scores differ"
        sel_off = comb_off = 0
        groups, same_order = N, True
        for s in range(nstage):
            ref_scores, ref_short, keep = z[f"scores{s}"][b], z[f"short{s}"][b], keeps[s]
            scale = max(scale, np.abs(ref_scores).max())
            if s > 0:
                kin = lad[s - 1][0]
                groups //= 2
                mine = t["comb"][comb_off:comb_off + groups * kin * kin].reshape(groups, kin * kin)
                comb_off += groups * kin * kin
                if same_order:      # candidate positions a*K'+b only line up while the shortlists had one order
                    assert _close(mine, ref_scores, scale), f"{name}[{b}]: combined scores of stage {s} differ"
            pos = t["sel_pos"][sel_off:sel_off + groups * keep].reshape(groups, keep)
            val = t["sel_val"][sel_off:sel_off + groups * keep].reshape(groups, keep)
            sel_off += groups * keep
            if not same_order:
                continue
            if np.array_equal(pos, ref_short):
                assert _close(val, np.take_along_axis(ref_scores, ref_short, axis=1), scale)
                continue
            # a different shortlist is only legitimate across a near-tie in the reference's own scores
            for g in range(groups):
                moved = np.concatenate([np.setxor1d(pos[g], ref_short[g]), ref_short[g][pos[g] != ref_short[g]],
                                        pos[g][pos[g] != ref_short[g]]])
                if moved.size:
                    v = ref_scores[g][moved]
                    assert v.max() - v.min() <= 4 * RTOL * scale, f"{name}[{b}] stage {s} group {g}: shortlist differs"
            same_order = False
            reordered += 1
        if same_order:
            assert np.array_equal(t["idx"], z["idx_out"][b]), f"{name}[{b}]: result of the pass differs"
    assert reordered <= 1, f"{name}: {reordered} of {int(z['nvec'])} vectors hit a near-tie reordering"
