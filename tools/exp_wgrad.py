"""Weight gradient of the trainer's second phase (dW = s G^T x, 2,048 x 512 from 4,096 rows): the bf16-piece kernel against the
fp32 MFMA kernel (MCQ_WGRAD_F32=1), time and error against the fp64 product.  python tools/exp_wgrad.py   (needs the GPU)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
for (B, M, D) in ((4096, 2048, 512), (4096, 1024, 256), (512, 2048, 512), (4000, 2048, 1024)):
    torch.manual_seed(1)
    G = torch.randn(B, M, device=dev) * 0.01
    x = torch.randn(B, D, device=dev)
    s = torch.tensor([1.0], device=dev)
    gW = torch.empty(M, D, device=dev)
    gb = torch.empty(M, device=dev)
    ws = torch.empty(L.mcq_weight_grad_workspace_bytes(B, M, D), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: L.mcq_weight_grad(G.data_ptr(), x.data_ptr(), B, M, D, s.data_ptr(), gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel(), st)
    for _ in range(5):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    ref = G.double().t() @ x.double()
    err = float((gW.double() - ref).abs().max() / ref.abs().max())
    errb = float((gb.double() - G.double().sum(0)).abs().max() / G.double().sum(0).abs().max())
    f32 = float(((G.t() @ x).double() - ref).abs().max() / ref.abs().max())
    print(f"B={B} M={M} D={D}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, max error / max |dW| {err:.2e} (rocBLAS fp32: {f32:.2e}), db {errb:.2e}",
          flush=True)
