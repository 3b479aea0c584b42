"""Sustained decode time (50 launches after 50 untimed, HIP events) per shape, for each value of a tuning hook given as NAME=v1,v2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
from quantization_amd._lib import lib
L = lib()
name, vals = (sys.argv[1].split("=") + [""])[:2] if len(sys.argv) > 1 else ("MCQ_DECODE_BLK", "1,0")
vals = vals.split(",")
for (D, K, N, B) in [(512, 256, 8, 65536), (512, 256, 8, 1048576), (256, 256, 4, 65536), (256, 256, 4, 1048576), (512, 256, 4, 65536), (1024, 256, 16, 65536), (1024, 256, 16, 262144), (512, 256, 16, 65536), (512, 64, 8, 65536)]:
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in gen.synthetic_state(7, D, K, N).items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    with torch.no_grad():
        codes = torch.randint(0, K, (B, N), device="cuda", dtype=torch.uint8)
        os.environ.pop(name, None)
        y0 = q.decode(codes).clone()
        blob = q._prepared(any_flavour=True)
        st = torch.cuda.current_stream().cuda_stream
        res = []
        for _ in range(300): L.mcq_decode(codes.data_ptr(), 1, N, min(B, 65536), blob.data_ptr(), N, K, D, y0.data_ptr(), st)     # clocks up
        for v in vals:
            os.environ[name] = v
            y = torch.empty_like(y0)
            reps = 50 if B <= 65536 else 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(reps): L.mcq_decode(codes.data_ptr(), 1, N, B, blob.data_ptr(), N, K, D, y.data_ptr(), st)
            e0.record()
            for _ in range(reps): L.mcq_decode(codes.data_ptr(), 1, N, B, blob.data_ptr(), N, K, D, y.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res.append(f"{name}={v}: {ms*1e3:8.1f} us {B*(N+4*D)/ms/1e9:6.2f} TB/s same={torch.equal(y, y0)}")
        os.environ.pop(name, None)
    print(f"D={D} K={K} N={N} B={B}: " + " | ".join(res), flush=True)
    del q, codes, y0, y
    torch.cuda.empty_cache()
