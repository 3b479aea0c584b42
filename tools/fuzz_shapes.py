"""Random-shape differential run of the HIP path against the CPU oracle (encode with 0..3 passes, the logits hook, decode).
One-off confidence run for kernels with tile / padding logic (partial 128-row tiles, dims that are no multiple of 128, every
codebook size): python tools/fuzz_shapes.py [cases] [seed]   (needs the GPU; the permanent versions are tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from quantization_amd import Quantizer, synthetic as gen  # noqa: E402
from oracle.oracle import OracleQuantizer  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    K = int(rs.choice([16, 32, 64, 128, 256, 512, 1024]))
    N = int(rs.choice([1, 2, 4, 8, 16, 32, 64]))
    while N * K > 16384:        # the supported domain: a Gram matrix of at most 16,384 rows
        N //= 2
    if K == 16 and N == 1:      # (bytes of one 16-entry codebook: the reference's own packing yields an empty tensor)
        N = 2
    D = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(300, 1100)]))
    B = int(rs.choice([rs.randint(1, 130), rs.randint(130, 700), rs.randint(700, 3000)]))
    if N * K * N * K * 4 > 300e6 or N >= 32 and B > 600:
        B = min(B, 300)
    if N * K >= 8192:             # (a Gram matrix of 0.27-1 GB; the CPU oracle's table is the slow side)
        B, D = min(B, 96), min(D, 200)
    sd = gen.synthetic_state(1000 + c, D, K, N)
    q = Quantizer(D, K, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    o = OracleQuantizer(sd["centers"], float(sd["centers_scale"]), sd["to_logits.weight"], sd["to_logits.bias"],
                        float(sd["logits_scale"]))
    x = gen.make_gaussian(2000 + c, B, D)
    if c % 3 == 1:      # rows of very different magnitude, a zero row
        x *= np.exp(rs.uniform(-20, 20, size=(B, 1))).astype(np.float32)
        x[rs.randint(0, B)] = 0
    xg = torch.from_numpy(x).cuda()
    ok = True
    with torch.no_grad():
        lg = q.logits_kernel(xg[:64]).cpu().numpy()
        ok &= np.array_equal(lg.view(np.uint32), o.logits(x[:64]).view(np.uint32))
        for it in (0, 1, 3):
            got = q.encode(xg, it, as_bytes=False).cpu().numpy()
            ok &= np.array_equal(got, o.compute_indexes(x, it))
        codes = o.encode(x, 2, as_bytes=K <= 256)
        ok &= np.array_equal(q.decode(torch.from_numpy(codes).cuda()).cpu().numpy(), o.decode(codes))
    print(f"case {c}: D={D} K={K} N={N} B={B} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del q, xg
    torch.cuda.empty_cache()
print("cases", cases, "mismatching", bad)
sys.exit(1 if bad else 0)
