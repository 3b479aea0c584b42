"""Times encode_from_host against the device-resident encode for several chunk sizes."""
import time, torch, sys
import numpy as np
sys.path.insert(0, ".")
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
D, N, K, B = 512, 8, 256, 131072
q = Quantizer(D, K, N)
sd = q.state_dict()
for k, v in gen.synthetic_state(3, D, K, N).items():
    sd[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(sd)
q = q.cuda()
with torch.no_grad():
    xh = torch.from_numpy(gen.make_gaussian(4, B, D)).pin_memory()
    xd = xh.cuda()
    for _ in range(2):
        q.encode(xd[:65536], 5)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        q.encode(xd[:65536], 5); q.encode(xd[65536:], 5)
    torch.cuda.synchronize(); dev = (time.perf_counter() - t) / 3
    print("device-resident  %.2f ms  %.2f M/s" % (dev * 1e3, B / dev / 1e6))
    for chunk in (65536, 32768, 16384, 8192):
        q.encode_from_host(xh, 5, chunk=chunk)
        t = time.perf_counter()
        for _ in range(3):
            q.encode_from_host(xh, 5, chunk=chunk)
        h = (time.perf_counter() - t) / 3
        print("host chunk %6d  %.2f ms  %.2f M/s" % (chunk, h * 1e3, B / h / 1e6))
    t = time.perf_counter(); xd2 = xh.cuda(non_blocking=True); torch.cuda.synchronize()
    print("plain H2D of the batch %.2f ms" % ((time.perf_counter() - t) * 1e3))
