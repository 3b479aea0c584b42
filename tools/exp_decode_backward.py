"""mcq_decode_backward_u8 time against the batch size (is the scatter bound by a latency chain or by bytes?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd._lib import lib
L = lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for (N, K, D) in [(16, 16, 512), (8, 256, 512), (4, 256, 256)]:
    for B in (512, 1024, 2048, 4096, 8192, 16384, 65536):
        g = torch.randn(B, D, device=dev)
        codes = torch.randint(0, K, (B, N), device=dev, dtype=torch.uint8)
        out = torch.empty(N, K, D, device=dev)
        for _ in range(5): L.mcq_decode_backward_u8(g.data_ptr(), codes.data_ptr(), B, N, K, D, out.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.mcq_decode_backward_u8(g.data_ptr(), codes.data_ptr(), B, N, K, D, out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"N={N} K={K} D={D} B={B}: {us:8.1f} us   {B * N * D * 4 / us / 1e6:7.2f} TB/s of rows read   {us / B * 1e3:6.2f} ns per vector", flush=True)
