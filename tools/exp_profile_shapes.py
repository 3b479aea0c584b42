"""Per-launch-category times (mcq_profile_encode: the shipped launch sequence) of the other BASELINE shapes."""
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
dev = torch.device("cuda:0")
shapes = [(256, 4, 256, 65536), (1024, 16, 256, 65536), (512, 16, 16, 4096), (512, 8, 256, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for D, N, K, B in shapes:
    q = load_quantizer(gen.synthetic_state(103, D, K, N), D, K, N, dev)
    x = torch.randn(B, D, device=dev)
    with torch.no_grad():
        q.encode(x, 5)
        blob = q._prepared()
    ws = q._workspace(B, dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ms = (ctypes.c_float * 32)()
    cnt = (ctypes.c_int * 32)()
    acc = np.zeros(32)
    for _ in range(3):
        n = L.mcq_profile_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, 5, ws.data_ptr(), ws.numel(), st, ms, cnt, 32)
        assert n > 0
        acc[:n] += np.array(ms[:n]) / 3
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(5):
        q.encode(x, 5)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 5 * 1e3
    print(f"dim {D}, {N} x {K}, {B} vectors: encode {wall:.3f} ms, sum of launches {acc.sum():.3f} ms")
    for i in range(n):
        if cnt[i]:
            print(f"   {L.mcq_profile_category_name(i).decode():28s} {cnt[i]:3d} x {acc[i] / cnt[i]:8.4f} ms = {acc[i]:8.3f} ms")
