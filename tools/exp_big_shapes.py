import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer
from oracle.oracle import OracleQuantizer
for (D, K, N, B, it) in [(1024, 16, 64, 48, 1), (768, 256, 32, 48, 1), (2048, 256, 8, 100, 2), (4096, 256, 4, 70, 2), (1000, 128, 16, 64, 2), (520, 64, 2, 200, 3)]:
    st = gen.synthetic_state(500 + D + N, D, K, N)
    q = Quantizer(D, K, N); sd = q.state_dict()
    for k, v in st.items(): sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); q = q.cuda()
    o = OracleQuantizer(st["centers"], float(st["centers_scale"]), st["to_logits.weight"], st["to_logits.bias"], float(st["logits_scale"]))
    x = gen.make_gaussian(7 + D, B, D)
    with torch.no_grad():
        got = q.encode(torch.from_numpy(x).cuda(), it, as_bytes=False).cpu().numpy()
        q.skip_fixed_points = True
        got2 = q.encode(torch.from_numpy(x).cuda(), it, as_bytes=False).cpu().numpy()
        dec = q.decode(torch.from_numpy(got).cuda()).cpu().numpy()
    want = o.compute_indexes(x, it)
    print((D, K, N, B, it), "codes equal:", np.array_equal(got, want), "skip equal:", np.array_equal(got2, want), "decode equal:", np.array_equal(dec, o.decode(want.astype(np.uint8))), flush=True)
