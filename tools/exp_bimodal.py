"""Is the GEMM time bimodal across processes / allocations?  Runs the stage-0 GEMM repeatedly with
fresh allocations and reports per-kernel times and buffer addresses."""
import sys, os, ctypes, time
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
L = _lib.lib()
blob = q._prepared()
ms = (ctypes.c_float * 32)()
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    x = torch.randn(B, D, device="cuda")
    ws = torch.empty(L.mcq_encode_workspace_bytes(B, N, K, D) + trial * 4096 * 7, dtype=torch.uint8, device="cuda")
    keep.append(torch.empty(1 << (20 + trial % 5), dtype=torch.uint8, device="cuda"))   # perturb the allocator
    st = torch.cuda.current_stream().cuda_stream
    L.mcq_profile_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, 5, ws.data_ptr(), ws.numel(), st, ms, 32)
    n = L.mcq_profile_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, 5, ws.data_ptr(), ws.numel(), st, ms, 32)
    print(f"trial {trial}: x={x.data_ptr():#x} ws={ws.data_ptr():#x} logits={ms[0]:.3f} stage0={ms[2]/5:.3f} pair1={ms[4]/5:.3f} pair3={ms[6]/5:.3f}", flush=True)
