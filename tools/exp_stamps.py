"""Where a wave's time goes: phase stamps (s_memtime) of sampled waves of the three large pass kernels.

Needs the debug build of the library (built here if missing):
    hipcc ... -DMCQ_STAMPS -o quantization_amd/lib/libmcq_stamps.so      (python __graft_entry__.py --stamps)
run as   MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=quantization_amd/lib/libmcq_stamps.so python tools/exp_stamps.py
The stamps wait for everything outstanding (s_waitcnt 0) before they read the clock, so the build is a few per cent slower
than the product; it returns the same codes.  Output: per kernel the median / p90 clocks of each phase and of a wave's life, the
number of sampled waves alive at a time, and the gap between one sampled wave's end and the next start on... (not available: the
stamps carry no CU id)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import _lib, synthetic as gen  # noqa: E402
from bench import load_quantizer  # noqa: E402

L = _lib.lib()
assert hasattr(L, "mcq_debug_stamps"), "load the -DMCQ_STAMPS build through MCQ_LIB_PATH"
L.mcq_debug_stamps.restype = ctypes.c_int
L.mcq_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
D, N, K, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 8, 256, 65536)
dev = torch.device("cuda:0")
q = load_quantizer(gen.synthetic_state(103, D, K, N), D, K, N, dev)
x = torch.randn(B, D, device=dev)
with torch.no_grad():
    q.encode(x, 5)
    q.encode(x, 5)
KERNELS, WAVES, SLOTS = 4, 1 << 14, 8
buf = np.zeros(KERNELS * WAVES * SLOTS, np.uint64)
assert L.mcq_debug_stamps(None, 0, 1) == SLOTS           # clear
with torch.no_grad():
    q.encode(x, 1)                                        # ONE pass: one launch of each kernel fills the sample slots
assert L.mcq_debug_stamps(buf.ctypes.data, buf.size, 0) == SLOTS
buf = buf.reshape(KERNELS, WAVES, SLOTS).astype(np.int64)
names = ["k_tf_stage0  (every 16th workgroup, wave 0)", "k_tf_pair0   (every 64th wave)", "k_tf_level1: sibling combine (every 64th)",
         "k_tf_level1: cousin table (every 64th)"]
phases = [["rows + x.C in, scores formed", "selection", "-> end"],
          ["lists in", "leaf gathered, scores formed", "selection, list written", "-> end"],
          ["list bytes in", "masks, all gathers back", "tables assembled + summed", "selection, list written", "-> end"],
          ["list bytes in", "masks, all gathers back", "tables assembled + summed", "table stored", "-> end"]]
for k in range(KERNELS):
    s = buf[k]
    s = s[s[:, 0] > 0]
    if not len(s):
        continue
    marks = [0] + [i for i in range(1, SLOTS - 1) if (s[:, i] > 0).all()] + [SLOTS - 1]
    life = s[:, SLOTS - 1] - s[:, 0]
    t0, t1 = s[:, 0].min(), s[:, SLOTS - 1].max()
    print(f"== {names[k]}: {len(s)} sampled waves; kernel span {t1 - t0} clocks (s_memtime ticks)")
    print(f"   life of a wave: median {int(np.median(life))}, p10 {int(np.percentile(life, 10))}, p90 {int(np.percentile(life, 90))}")
    for a, b_, nm in zip(marks[:-1], marks[1:], phases[k]):
        d = s[:, b_] - s[:, a]
        print(f"   {nm:34s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}  ({np.median(d) / np.median(life) * 100:4.1f} % of the median life)")
    # sampled waves alive at a time (x the sampling factor = waves alive on the chip)
    ev = np.concatenate([np.stack([s[:, 0], np.ones(len(s), np.int64)], 1), np.stack([s[:, SLOTS - 1], -np.ones(len(s), np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    alive = np.cumsum(ev[:, 1])
    mid = alive[len(alive) // 4: 3 * len(alive) // 4]
    print(f"   sampled waves alive (middle half of the launch): mean {mid.mean():.1f}")
    starts = np.sort(s[:, 0])
    print(f"   sampled-wave start rate: one per {np.median(np.diff(starts)):.1f} clocks")
