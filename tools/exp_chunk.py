"""Encode time of the headline shape against the chunk the batch is cut into (the workspace the caller hands over decides it):
does a chunk whose x.C products (8 KB per vector) stay in the 256 MB memory-side cache across the five passes pay?"""
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
L = _lib.lib(); blob = q._prepared()
full = L.mcq_encode_workspace_bytes(B, N, K, D)
ref = None
for chunk in (65536, 32768, 16384, 8192, 4096, 2048):
    nbytes = L.mcq_encode_workspace_bytes(chunk, N, K, D)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(B, N, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def once():
        rc = L.mcq_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, 5, out.data_ptr(), None, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    for _ in range(3): once()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    if ref is None: ref = out.clone()
    print(f"chunk {chunk:6d} ({nbytes / 2**20:7.1f} MB of workspace): {dt * 1e3:.3f} ms  {B / dt / 1e6:.2f} M vectors/s  same={bool(torch.equal(ref, out))}", flush=True)
