"""Experiment: does running two half-batches on two streams (GEMM-bound and gather-bound kernels
co-resident) raise encode throughput?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer, _lib

D, N, K, B = 512, 8, 256, 65536
state = gen.synthetic_state(103, D, K, N)
q = Quantizer(D, K, N); sd = q.state_dict()
for k, v in state.items(): sd[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(sd); q = q.cuda()
x = torch.randn(B, D, device="cuda")
L = _lib.lib()
blob = q._prepared()

def run(nsplit, reps=3):
    parts = [x[i * B // nsplit:(i + 1) * B // nsplit].contiguous() for i in range(nsplit)]
    wss = [torch.empty(L.mcq_encode_workspace_bytes(p.shape[0], N, K, D), dtype=torch.uint8, device="cuda") for p in parts]
    outs = [torch.empty(p.shape[0], N, dtype=torch.uint8, device="cuda") for p in parts]
    streams = [torch.cuda.Stream() for _ in parts]
    def once():
        for p, ws, o, s in zip(parts, wss, outs, streams):
            rc = L.mcq_encode(p.data_ptr(), p.shape[0], blob.data_ptr(), q._lscale_exp, N, K, D, 5, o.data_ptr(), None,
                              ws.data_ptr(), ws.numel(), s.cuda_stream)
            assert rc == 0
    once(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return dt, torch.cat(outs)

ref = None
for ns in (1, 2, 4, 8):
    dt, codes = run(ns)
    if ref is None: ref = codes
    print(f"streams={ns}: {dt*1e3:.2f} ms  {B/dt/1e6:.3f} M vec/s  same={bool(torch.equal(ref, codes))}")
