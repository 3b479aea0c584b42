"""Gate run for the CENTERED table form (oracle/mcq_oracle.c, "CENTERING"): the oracle's codes against the reference's on
every fixture, with the codebook means taken out of the tables (default) and without (MCQ_ORACLE_CENTER=0).

    python tools/exp_gram/gate_center.py [fixture prefix]
    MCQ_ORACLE_CENTER=0 python tools/exp_gram/gate_center.py

CPU only (the oracle and the committed fixtures); results of round 4 in results_r04_center.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import fixtures  # noqa: E402
from oracle.oracle import OracleQuantizer  # noqa: E402

tot = tothard = cases = 0
for name in fixtures.names(sys.argv[1] if len(sys.argv) > 1 else ""):
    fx = fixtures.load(name)
    s = fx["state"]
    o = OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"], float(s["logits_scale"]))
    for it in fx["iters"]:
        codes = o.compute_indexes(fx["x"], it)
        ref, margin = fx[f"codes_it{it}"], fx[f"margin_it{it}"]
        bad = (codes != ref).any(axis=1)
        hard = bad & (margin >= fixtures.NEAR_TIE)
        tot += int(bad.sum())
        tothard += int(hard.sum())
        cases += len(ref)
        if bad.any():
            print(f"{name} iters={it}: oracle != reference on {int(bad.sum())} vectors ({int(hard.sum())} with a clear margin); "
                  f"reference vs its own permuted run: {int(fx.get(f'reorder_noise_it{it}', -1))}; "
                  f"largest fp64 margin among them {float(margin[bad].max()):.2e}")
print(f"centering {'off' if os.environ.get('MCQ_ORACLE_CENTER', '1')[0] == '0' else 'on'}: "
      f"{tot} differing of {cases} (vector x pass-count) cases, {tothard} with a clear margin")
