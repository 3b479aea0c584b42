"""Generates tests/golden/trace_*.npz: intermediates of ONE Quantizer._refine_indexes pass of the
reference (quantization/quantization.py:308-547) for 8 vectors per stored quantizer state, so that a
divergence can be localised to a stage (fixture F3).  Runs only in the build container.

    python tests/golden/make_golden_trace.py

The reference's function is one block, so the intermediates are captured by observing the
torch.sort calls it makes (:474): the argument of the i-th call is the candidate scores entering the
i-th prune (stage-0 `cur_sumsq` (B,N,K) first, then every combined table (B,N',K'^2)), and the first
`keep` sorted positions are the shortlist.  Stored per fixture: the starting indexes (the reference's
own argmax initialisation), every score table, every shortlist, and the pass's result.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402  (imports the reference with the h5py stub)
from golden import fixtures  # noqa: E402

STATES = ["trained_d64_b4_p1", "trained_d64_b4_p2", "trained_d64_b8_p1", "trained_d64_b8_p2", "synth_d40_k64_n8", "synth_d32_k16_n64"]
NVEC = 8


def trace_one(name):
    fx = fixtures.load(name)
    D, K, N = fx["D"], fx["K"], fx["N"]
    q = mg.ref_quantizer(fx["state"], D, K, N)
    x = torch.from_numpy(fx["x"][:NVEC])
    calls = []
    real_sort = torch.sort

    def spy(inp, *a, **kw):
        r = real_sort(inp, *a, **kw)
        calls.append((inp.detach().clone().numpy(), r[1].detach().clone().numpy()))
        return r

    with torch.no_grad():
        idx0 = q._compute_indexes(x, 0)
        torch.sort = spy
        try:
            idx1 = q._refine_indexes(x, idx0)
        finally:
            torch.sort = real_sort
    first, lad = mg_ladder(N, K)
    keeps = [first] + [ko for (_, ko) in lad]
    assert len(calls) == len(keeps), (len(calls), keeps)
    out = dict(D=D, K=K, N=N, source=name, nvec=NVEC, idx_in=idx0.numpy().astype(np.uint8),
               idx_out=idx1.numpy().astype(np.uint8), keeps=np.asarray(keeps, np.int32))
    for i, ((scores, order), keep) in enumerate(zip(calls, keeps)):
        out[f"scores{i}"] = scores.astype(np.float32)                 # (B, N', K')
        out[f"short{i}"] = order[:, :, :keep].astype(np.int32)        # (B, N', keep), best first
    np.savez_compressed(os.path.join(HERE, "trace_" + name + ".npz"), **out)
    print(name, "stages", [(c[0].shape, k) for c, k in zip(calls, keeps)])


def mg_ladder(N, K):
    """(first_keep, [(Kin, Kout)...]) by the reference's rule :453-463 / :465-547."""
    lad, first = [], None
    n, k, L = N, K, 1
    while True:
        kc = mg.k_cutoff(K, L)
        if n == 1 and k == 1:
            break
        if k > kc or n == 1:
            keep = 1 if n == 1 else kc
            if first is None:
                first = keep
            else:
                lad[-1] = (lad[-1][0], keep)
            k = keep
        else:
            if first is None:          # K <= cutoff at stage 0: nothing pruned before the first combine
                first = k
            lad.append((k, None))
            n, k, L = n // 2, k * k, L * 2
    return first, lad


if __name__ == "__main__":
    for nm in STATES:
        trace_one(nm)
