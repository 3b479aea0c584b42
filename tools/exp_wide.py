"""Encode time of codebooks of 512 / 1,024 entries (two-byte entries, as_bytes=False) beside the 8 x 256 headline shape:
python tools/exp_wide.py   (needs the GPU)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import Quantizer, synthetic as gen  # noqa: E402

B, D = 65536, 512
x = torch.from_numpy(gen.make_gaussian(5, B, D)).cuda()
for K, N in ((256, 8), (512, 8), (1024, 8), (1024, 4), (512, 16)):
    sd = gen.synthetic_state(10 + K + N, D, K, N)
    q = Quantizer(D, K, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    for _ in range(3):
        q.encode(x, 5, as_bytes=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        q.encode(x, 5, as_bytes=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"K={K} N={N} D={D} B={B}: {ms:.2f} ms per encode of 5 passes, {B / ms / 1e3:.2f} M vectors/s, "
          f"{N * int(np.log2(K))} bits per vector", flush=True)
    del q
    torch.cuda.empty_cache()
