"""Gate of the fixed-point GEMM definition (oracle/mcq_oracle.c "fixdot"): codes of the oracle against the codes the
reference returned, over every fixture and refinement count.  Run from the repo root:  python tools/exp_fixdot/count_mismatches.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import fixtures  # noqa: E402
from oracle.oracle import OracleQuantizer  # noqa: E402

tot = bad_t = hard_t = 0
for n in fixtures.names():
    fx = fixtures.load(n)
    s = fx["state"]
    o = OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"],
                        float(s["logits_scale"]))
    for it in fx["iters"]:
        c = np.asarray(o.compute_indexes(fx["x"], it)).reshape(fx[f"codes_it{it}"].shape)
        bad = (c != fx[f"codes_it{it}"]).any(axis=1)
        m = fx[f"margin_it{it}"]
        tot += len(bad)
        bad_t += int(bad.sum())
        hard_t += int((bad & (m >= fixtures.NEAR_TIE)).sum())
        if bad.any():
            print(n, "iters", it, "rows", np.flatnonzero(bad), "fp64 margins", m[bad])
print("cases", tot, "mismatches", bad_t, "with a clear margin", hard_t)
