"""mcq_weight_grad (split-K fp32 MFMA) against torch.mm (rocBLAS) on the trainer's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import _lib
L = _lib.lib()
dev = torch.device("cuda")
for (B, M, D) in [(4096, 2048, 512), (4096, 256, 512), (600, 1024, 256), (65536, 2048, 512)]:
    G = torch.randn(B, M, device=dev) * 0.01
    x = torch.randn(B, D, device=dev)
    s = torch.ones(1, device=dev)
    gW = torch.empty(M, D, device=dev); gb = torch.empty(M, device=dev)
    ws = torch.empty(L.mcq_weight_grad_workspace_bytes(B, M, D), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def mine():
        assert L.mcq_weight_grad(G.data_ptr(), x.data_ptr(), B, M, D, s.data_ptr(), gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel(), st) == 0
    def lib():
        return torch.mm(G.t(), x), G.sum(0)
    for name, fn in (("mcq_weight_grad", mine), ("torch.mm + sum", lib)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"B={B} M={M} D={D} {name:16s} {us:8.1f} us  {2.0*B*M*D/us/1e6:6.1f} TFLOP/s", flush=True)
    ref = torch.mm(G.double().t(), x.double())
    print("   max rel err", float((gW.double() - ref).abs().max() / ref.abs().max()))
