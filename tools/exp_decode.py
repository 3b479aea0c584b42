"""Decode throughput (vectors/s, output GB/s) for several shapes; MCQ_DECODE_SLICED=0 gives the per-vector kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
for (D, K, N, B) in [(512, 256, 8, 65536), (512, 256, 8, 1048576), (256, 256, 4, 1048576), (1024, 256, 16, 262144), (512, 16, 16, 1048576), (40, 64, 8, 1048576)]:
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in gen.synthetic_state(7, D, K, N).items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    with torch.no_grad():
        codes = torch.randint(0, K, (B, N), device="cuda", dtype=torch.uint8)
        y = q.decode(codes); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): q.decode(codes)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        ref = q.get_centers()[torch.arange(N, device="cuda"), codes[:4096].long()]      # (4096, N, D)
        acc = ref[:, 0]
        for n in range(1, N): acc = acc + ref[:, n]
        ok = torch.equal(acc, y[:4096])
    print(f"D={D} K={K} N={N} B={B}: {dt*1e3:.3f} ms  {B/dt/1e9:.3f} Gvec/s  {B*(N+4*D)/dt/1e9:.0f} GB/s out  bit-exact vs torch gather-sum: {ok}")
