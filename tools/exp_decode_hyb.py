"""Decode kernels at the BASELINE shapes: what mcq_decode picks (the pipelined LDS-resident kernel; 8 + 8 hybrid for 16 x 256)
against the XCD-sliced kernel (MCQ_DECODE_LDS_MIN set beyond the batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
for (D, K, N, B) in [(512, 256, 8, 65536), (512, 256, 8, 16384), (1024, 256, 16, 65536), (1024, 256, 16, 16384), (256, 256, 8, 65536),
                     (512, 128, 16, 65536), (512, 256, 8, 1048576)]:
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in gen.synthetic_state(7, D, K, N).items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    with torch.no_grad():
        codes = torch.randint(0, K, (B, N), device="cuda", dtype=torch.uint8)
        ref = None
        for name, env in (("sliced", str(1 << 40)), ("default", None)):
            if env is None: os.environ.pop("MCQ_DECODE_LDS_MIN", None)
            else: os.environ["MCQ_DECODE_LDS_MIN"] = env
            for _ in range(3): y = q.decode(codes)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): q.decode(codes)
            e1.record(); torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) / 20 * 1e-3
            ref = y if ref is None else ref
            print(f"D={D} K={K} N={N} B={B} {name:8s}: {dt*1e6:8.1f} us  {B*(N+4*D)/dt/1e9:7.0f} GB/s  equal: {torch.equal(y, ref)}", flush=True)
    del q, codes, y, ref
    torch.cuda.empty_cache()
