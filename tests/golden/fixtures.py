"""Loads a golden fixture (see make_golden.py) and regenerates its inputs."""
import glob
import os

import numpy as np

from . import gen

HERE = os.path.dirname(os.path.abspath(__file__))

# a vector whose fp64 decision margin is below this is a near-tie: the reference's
# own codes for it depend on fp32 summation order.  Round 6: the margin is `margin2_it*`, the smallest decision
# gap along the fp64 search relative to the two COMPETING SCORES (fp64_search._gap2); the older `margin_it*`, relative to |x|^2 + E,
# flagged every row of a fixture with offset frames (2,048 of 2,048), so that "no difference with a clear margin" could not fail there
NEAR_TIE = 2e-6


def names(prefix=""):
    """encode/decode fixtures (the trainer trajectory fixture has its own loader)"""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, prefix + "*.npz")))
                  if not n.startswith(("trainer_", "trace_", "jcl_", "downstream_", "loss_")))


def trace_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "trace_*.npz")))


class PinnedState(dict):
    """a quantizer state (the state-dict entries) that also carries, as an ATTRIBUTE, the fp32 scale factors the reference's run
    computed with (`scales_exp` of the fixture): the helpers that build a Quantizer / an OracleQuantizer / the fp64 search from a
    state pin them, so the problem instance is the reference's on every host (torch's fp32 exp differs in the last bit between CPUs)"""
    scales_exp = None


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    fx = {k: z[k] for k in z.files}
    D, K, N, B = int(fx["D"]), int(fx["K"]), int(fx["N"]), int(fx["B"])
    if "state.centers" in fx:
        state = {k[len("state."):]: fx[k] for k in fx if k.startswith("state.")}
    else:
        state = gen.synthetic_state(int(fx["state_seed"]), D, K, N)
        assert gen.checksum(state["centers"]) == float(fx["centers_checksum"]), "regenerated state differs"
    x = gen.make_kind(str(fx["x_kind"]), int(fx["x_seed"]), B, D)
    assert gen.checksum(x) == float(fx["x_checksum"]), "regenerated input differs from the fixture's"
    state = PinnedState(state)
    if "scales_exp" in fx:
        state.scales_exp = (float(fx["scales_exp"][0]), float(fx["scales_exp"][1]))
    fx.update(D=D, K=K, N=N, B=B, state=state, x=x)
    fx["iters"] = sorted(int(k[len("codes_it"):]) for k in fx if k.startswith("codes_it"))
    return fx


def check_codes(fx, it, codes, what, certify=True):
    """codes must equal the reference's wherever the decision margin is not a near-tie; every row that differs is CERTIFIED
    (certify.certify_row: the pass and the node of the combine tree where the two part, the fp64 gap there below 1e-6 of the
    competing scores, the reconstruction errors of the two results within 2 %)"""
    ref = fx[f"codes_it{it}"]
    margin = fx[f"margin2_it{it}"] if f"margin2_it{it}" in fx else fx[f"margin_it{it}"]
    codes = np.asarray(codes).reshape(ref.shape)
    bad = (codes != ref).any(axis=1)
    hard = bad & (margin >= NEAR_TIE)
    assert not hard.any(), (f"{what}: {int(hard.sum())} vectors differ from the reference with a clear margin, "
                            f"first at {np.flatnonzero(hard)[:5]}")
    # near-tie differences must stay rare, or the comparison means nothing.  Where the fixture holds the reference's OWN
    # reorder noise (its codes against those of its feature-permuted run, make_golden.py::permuted_reference) that is the
    # yardstick: no more than ONE vector above it; otherwise a flat 0.05 %
    key = f"reorder_noise_it{it}"
    # (+ 1, and never more than 0.5 % of the rows whatever a regenerated fixture stores)
    limit = min(int(fx[key]) + 1, max(2, int(0.005 * len(ref)))) if key in fx else max(2, 0.0005 * len(ref))
    assert bad.sum() <= limit, f"{what}: {int(bad.sum())} near-tie differences of {len(ref)} (limit {limit})"
    if certify and bad.any() and "refpass" in fx:
        from . import certify as cert
        for row in np.flatnonzero(bad):
            c = cert.certify_row(fx, it, int(row), codes[row])
            print(f"{what}: row {row} certified -- {c['stage']}, gap {c['gap']:.2e}; SSE {c['sse_rel']:+.2e} of the reference's")
    return int(bad.sum())
