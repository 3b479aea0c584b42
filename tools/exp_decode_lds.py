"""LDS-resident decode against the XCD-sliced kernel at several batch sizes (MCQ_DECODE_LDS_MIN picks the kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
BIG = str(1 << 40)
for (D, K, N, B) in [(512, 256, 8, 65536), (512, 256, 8, 16384), (512, 256, 8, 1048576), (1024, 256, 16, 65536), (256, 256, 8, 65536)]:
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in gen.synthetic_state(7, D, K, N).items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    with torch.no_grad():
        codes = torch.randint(0, K, (B, N), device="cuda", dtype=torch.uint8)
        outs = {}
        for name, env in (("sliced", BIG), ("lds", "1")):
            os.environ["MCQ_DECODE_LDS_MIN"] = env
            y = q.decode(codes); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): q.decode(codes)
            e1.record(); torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) / 20 * 1e-3
            outs[name] = y
            print(f"D={D} N={N} B={B} {name:7s}: {dt*1e6:8.1f} us  {B*(N+4*D)/dt/1e9:7.0f} GB/s  equal to sliced: {torch.equal(y, outs['sliced'])}", flush=True)
