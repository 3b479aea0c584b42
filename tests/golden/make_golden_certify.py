"""Adds to every encode fixture (tests/golden/*.npz) what round 6's certification of mismatching rows needs.  Runs ONLY in the build
container (imports the reference through make_golden.py); the states are the ones the fixtures already hold, nothing is retrained.

    python tests/golden/make_golden_certify.py [name-prefix]

New fields per fixture (P = the largest stored pass count):
  refpass            (P + 1, B, N)  the REFERENCE's indexes after 0, 1, ..., P refinement passes (Quantizer.encode(x, p, as_bytes=False),
                                    quantization.py:244-275) -- so that the pass in which another implementation parts from it is known
  margin2_it{it}     (B,) float32   smallest decision gap along the fp64 search RELATIVE TO THE TWO COMPETING SCORES (fp64_search._gap2);
                                    margin_it{it}, normalised by |x|^2 + E, calls every row of an offset fixture a near-tie
  sse64_it{it}       (B,) float64   fp64 |sum_n C[n, code_n] - x|^2 of the reference's code: the outcome of its search
  scales_exp         (2,) float32   exp(10 centers_scale), exp(10 logits_scale) as the reference formed them on the generating machine
                                    (torch's fp32 exp differs in the last bit between CPUs: tests pin these, fixtures.PinnedState)
Checks while it runs: refpass[it] equals the stored codes_it{it}; the fp64 margins of the old definition equal the stored ones."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402  (imports the reference with the h5py stub)
import fp64_search as f64  # noqa: E402
from golden import fixtures  # noqa: E402


def patch(name):
    path = os.path.join(HERE, name + ".npz")
    fx = fixtures.load(name)
    raw = dict(np.load(path))
    D, K, N, x, sd = fx["D"], fx["K"], fx["N"], fx["x"], fx["state"]
    q = mg.ref_quantizer(sd, D, K, N)
    P = max(fx["iters"])
    dt = np.uint8 if K <= 256 else np.uint16
    refpass = np.stack([mg.ref_encode(q, x, p, as_bytes=False).astype(dt) for p in range(P + 1)])
    for it in fx["iters"]:
        assert np.array_equal(refpass[it], fx[f"codes_it{it}"]), (name, it, "the reference no longer reproduces the stored codes")
        c64, m_old, m2 = f64.search_fp64(sd, x, it)
        assert np.allclose(m_old, fx[f"margin_it{it}"].astype(np.float64), rtol=1e-3, atol=1e-9), (name, it)
        raw[f"margin2_it{it}"] = m2.astype(np.float32)
        raw[f"sse64_it{it}"] = f64.sse_fp64(sd, x, refpass[it])
        print(f"[{name}] iters={it}: near-tie rows (< 2e-6) old normalisation {int((m_old < 2e-6).sum())}, new {int((m2 < 2e-6).sum())} of {len(x)};"
              f" fp64 codes differ from the reference's on {int((c64 != refpass[it]).any(axis=1).sum())}")
    # the fp32 scale factors the reference computed with HERE (its get_centers() / _logits(): torch's fp32 exp on this machine)
    import torch
    with torch.no_grad():
        raw["scales_exp"] = np.asarray([float((q.centers_scale * q.scale_speed).exp()), float((q.logits_scale * q.scale_speed).exp())], np.float32)
    raw["refpass"] = refpass
    np.savez_compressed(path, **raw)


if __name__ == "__main__":
    import torch
    torch.set_num_threads(8)
    pref = sys.argv[1] if len(sys.argv) > 1 else ""
    for nm in fixtures.names(pref):
        patch(nm)
