import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
D, N, K = 512, 8, 256
st = gen.synthetic_state(103, D, K, N)
q = Quantizer(D, K, N); sd = q.state_dict()
for k, v in st.items(): sd[k] = torch.from_numpy(np.asarray(v))
q.load_state_dict(sd); q = q.cuda()
with torch.no_grad():
    for B in (64, 256, 1024, 4096, 16384):
        x = torch.randn(B, D, device="cuda")
        for _ in range(3): q.encode(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): c = q.encode(x)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 20
        # captured in a hipGraph (static input/output buffers)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            q.encode(x)
            with torch.cuda.graph(g, stream=s):
                cg = q.encode(x)
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 20
        print(f"B={B}: eager {eager*1e3:.3f} ms ({B/eager/1e6:.3f} M/s), graph {graph*1e3:.3f} ms ({B/graph/1e6:.3f} M/s), same={bool(torch.equal(c, cg))}")
