"""Differential run of k_tf_pass16 (16 entries per codebook, 8 or 16 codebooks: all refinement passes of a call in one launch, Gram
matrix in LDS) against the CPU oracle on random dims / batch sizes / pass counts, then its time beside the separate launches
(MCQ_PASS16=0 in a second process).  python tools/fuzz_pass16.py [cases] [seed]   (needs the GPU)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from quantization_amd import Quantizer, synthetic as gen  # noqa: E402
from oracle.oracle import OracleQuantizer  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    K, N = 16, (8, 16)[c & 1]
    D = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(300, 700)]))
    B = int(rs.choice([rs.randint(1, 9), rs.randint(9, 700), rs.randint(700, 6000)]))
    sd = gen.synthetic_state(3000 + c, D, K, N)
    q = Quantizer(D, K, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    o = OracleQuantizer(sd["centers"], float(sd["centers_scale"]), sd["to_logits.weight"], sd["to_logits.bias"],
                        float(sd["logits_scale"]))
    x = gen.make_gaussian(4000 + c, B, D)
    if c % 3 == 1:
        x *= np.exp(rs.uniform(-20, 20, size=(B, 1))).astype(np.float32)
        x[rs.randint(0, B)] = 0
    xg = torch.from_numpy(x).cuda()
    ok = True
    with torch.no_grad():
        for it in (1, 2, 5):
            ok &= np.array_equal(q.encode(xg, it, as_bytes=False).cpu().numpy(), o.compute_indexes(x, it))
        ok &= np.array_equal(q.encode(xg, 2, as_bytes=True).cpu().numpy(), o.encode(x, 2, as_bytes=True))
    print(f"case {c}: D={D} K={K} N={N} B={B} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("cases", cases, "mismatching", bad)
for N in (8, 16):
    sd = gen.synthetic_state(7, 512, 16, N)
    q = Quantizer(512, 16, N)
    st = q.state_dict()
    for k, v in sd.items():
        st[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(st)
    q = q.cuda()
    xg = torch.from_numpy(gen.make_gaussian(8, 4096, 512)).cuda()
    with torch.no_grad():
        for it in (1, 2, 5):
            for _ in range(20):
                q.encode(xg, it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                q.encode(xg, it)
            torch.cuda.synchronize()
            print(f"{os.environ.get('AB_TAG', '')} N={N} B=4096 D=512 passes={it}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per encode", flush=True)
sys.exit(1 if bad else 0)
