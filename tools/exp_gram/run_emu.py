"""Counts, per evaluation mode, the vectors whose codes differ from the reference fixtures."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
O._SO = "/tmp/libgram_emu2.so"; O.build = lambda force=False: O._SO
from golden import fixtures
lib = O._load()
lib.mcq_emu_modes.argtypes = [ctypes.c_int, ctypes.c_int]
modes = [tuple(int(v) for v in m.split(",")) for m in (sys.argv[1:] or ["0,0", "1,0", "2,0", "1,1"])]
names = fixtures.names()
for pm, sm in modes:
    lib.mcq_emu_modes(pm, sm)
    tot = hard = near = cases = 0
    t0 = time.time()
    for name in names:
        fx = fixtures.load(name)
        if os.environ.get("SKIP_BIG") and name.startswith(("config_b", "config_d")): continue
        s = fx["state"]
        o = O.OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"], float(s["logits_scale"]))
        for it in fx["iters"]:
            codes = o.compute_indexes(fx["x"], it)
            ref = fx[f"codes_it{it}"]; margin = fx[f"margin_it{it}"]
            bad = (codes != ref).any(axis=1)
            h = int((bad & (margin >= fixtures.NEAR_TIE)).sum())
            if bad.sum(): print(f"   pair={pm} s0={sm} {name} it={it}: {int(bad.sum())} differ ({h} hard) rows {np.flatnonzero(bad)[:6]} margins {margin[bad][:6]}")
            tot += int(bad.sum()); hard += h; cases += len(ref)
    print(f"MODE pair={pm} s0={sm}: {tot} mismatches ({hard} with clear margin) of {cases} cases  [{time.time()-t0:.0f}s]", flush=True)
