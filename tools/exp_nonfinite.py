import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
for (D,K,N) in [(64,256,8),(64,16,16),(32,256,1)]:
    st = gen.synthetic_state(1, D, K, N)
    q = Quantizer(D, K, N); sd = q.state_dict()
    for k, v in st.items(): sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    x = torch.randn(300, D, device="cuda")
    x[3] = float("nan"); x[7, 5] = float("inf"); x[9] = -float("inf"); x[11] = 1e30; x[13] = 0.0
    with torch.no_grad():
        for skip in (False, True):
            q.skip_fixed_points = skip
            c = q.encode(x, 5, as_bytes=False)
            y = q.decode(c)
    torch.cuda.synchronize()
    print((D,K,N), "ok", int(c.min()), int(c.max()), bool((c[[0,1,2,4,5,6]] >= 0).all()))
