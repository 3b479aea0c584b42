"""Generates tests/golden/*.npz by importing the reference (danpovey/quantization)
from /root/reference.  Runs ONLY in the build container; the fixtures it writes
are data (inputs are regenerated from seeds by gen.py; expected outputs, trained
states and fp64 decision margins are stored).

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py trained    # subset

Every expected code array is what the reference's own Quantizer.encode returned
(torch CPU fp32, quantization/quantization.py:244-275).  `margin` is the smallest
relative decision gap met along the same search carried out in fp64 (argmax of
the logits :301, every sort-and-truncate :474-478, the final top-1); a vector
whose margin is below ~1e-5 is a near-tie: the reference's own codes for it
depend on fp32 summation order, so tests accept a difference there.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")
import quantization as refq  # noqa: E402

import gen  # noqa: E402


# --------------------------------------------------------------------------
# fp64 run of the same search, recording decision margins (batched numpy)
# --------------------------------------------------------------------------
def k_cutoff(K, L):
    kc = 8 if K <= 16 else 16
    while L >= 4:
        L //= 4
        kc *= 2
    return min(kc, 128)


def _gap(sorted_vals, keep, norm):
    """relative gap between the last kept and the first dropped candidate"""
    if sorted_vals.shape[-1] <= keep:
        return np.full(sorted_vals.shape[:-1], np.inf)
    return (sorted_vals[..., keep] - sorted_vals[..., keep - 1]) / norm


def search_fp64(sd, x, iters):
    """returns (codes (B,N) int64, margin (B,) float64)"""
    cs = float((torch.tensor(float(sd["centers_scale"]), dtype=torch.float32) * 10.0).exp())
    ls = float((torch.tensor(float(sd["logits_scale"]), dtype=torch.float32) * 10.0).exp())
    C32 = (np.float32(cs) * sd["centers"].astype(np.float32)).astype(np.float32)
    C = C32.astype(np.float64)
    N, K, D = C.shape
    W = sd["to_logits.weight"].astype(np.float64)
    bias = sd["to_logits.bias"].astype(np.float64)
    codes = np.zeros((x.shape[0], N), np.int64)
    margin = np.zeros(x.shape[0])
    step = max(1, min(x.shape[0], (1 << 27) // (N * 32 * D * 8) ))
    for lo in range(0, x.shape[0], step):
        xb32 = x[lo:lo + step]
        xb = xb32.astype(np.float64)
        sx = (np.float32(ls) * xb32).astype(np.float32).astype(np.float64)
        logits = (sx @ W.T + bias).reshape(-1, N, K)
        srt = np.sort(logits, axis=2)
        m = ((srt[..., -1] - srt[..., -2]) / (np.abs(srt).max(axis=2) + 1e-300)).min(axis=1)
        idx = logits.argmax(axis=2)
        for _ in range(iters):
            idx, mi = _refine_fp64(C, xb, idx)
            m = np.minimum(m, mi)
        codes[lo:lo + step] = idx
        margin[lo:lo + step] = m
    return codes, margin


def _refine_fp64(C, x, idx):
    N, K, D = C.shape
    B = x.shape[0]
    ar = np.arange(N)
    old = C[ar[None, :], idx]                      # (B,N,D)
    xerr = old.sum(axis=1) - x                     # (B,D)
    E = (xerr ** 2).sum(-1)                        # (B,)
    norm = ((x ** 2).sum(-1) + E + 1e-300)
    xrem = xerr[:, None, :] - old
    R = (xrem ** 2).sum(-1)                        # (B,N)
    Q = (C ** 2).sum(-1)                           # (N,K)
    X = np.einsum("nkd,bnd->bnk", C, xrem)
    S = (R[..., None] + Q[None]) + 2 * X           # (B,N,K)
    margin = np.full(B, np.inf)
    Ng, L = N, 1
    keep = 1 if Ng == 1 else k_cutoff(K, L)
    order = np.argsort(S, axis=2, kind="stable")
    Ss = np.take_along_axis(S, order, axis=2)
    margin = np.minimum(margin, _gap(Ss, keep, norm[:, None]).min(axis=1))
    sel = order[..., :keep]                        # (B,N,keep)
    curS = Ss[..., :keep]
    tuples = sel[..., None]                        # (B,N,keep,1)
    deltas = C[ar[None, :, None], sel] - old[:, :, None, :]   # (B,N,keep,D)
    Kg = keep
    while Ng > 1:
        de, do = deltas[:, 0::2], deltas[:, 1::2]
        newN = Ng // 2
        dots = np.einsum("bgad,bgcd->bgac", de, do)
        comb = (curS[:, 0::2, :, None] + curS[:, 1::2, None, :]) - E[:, None, None, None] + 2 * dots
        comb = comb.reshape(B, newN, Kg * Kg)
        L *= 2
        keep = 1 if newN == 1 else k_cutoff(K, L)
        order = np.argsort(comb, axis=2, kind="stable")
        Ss = np.take_along_axis(comb, order, axis=2)
        margin = np.minimum(margin, _gap(Ss, keep, norm[:, None]).min(axis=1))
        sel = order[..., :keep]
        curS = Ss[..., :keep]
        a, b = sel // Kg, sel % Kg
        te = np.take_along_axis(tuples[:, 0::2], a[..., None], axis=2)
        to = np.take_along_axis(tuples[:, 1::2], b[..., None], axis=2)
        tuples = np.concatenate([te, to], axis=3)
        deltas = (np.take_along_axis(de, a[..., None], axis=2) + np.take_along_axis(do, b[..., None], axis=2))
        Ng, Kg = newN, keep
    return tuples[:, 0, 0, :], margin


# --------------------------------------------------------------------------
def ref_quantizer(sd_np, D, K, N):
    q = refq.Quantizer(dim=D, codebook_size=K, num_codebooks=N)
    sd = q.state_dict()
    for k, v in sd_np.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd)
    return q


def ref_encode(q, x, iters, as_bytes, chunk=256):
    outs = []
    with torch.no_grad():
        for lo in range(0, x.shape[0], chunk):
            outs.append(q.encode(torch.from_numpy(x[lo:lo + chunk]), iters, as_bytes).numpy())
    return np.concatenate(outs, axis=0)


def np_state(q):
    return {k: v.detach().cpu().numpy().copy() for k, v in q.state_dict().items()}


def permuted_reference(sd, x, D, K, N, seed=4242):
    """The reference itself on the SAME problem with the feature axis permuted (inputs, centers and classifier columns alike):
    mathematically the same search, a different fp32 summation order inside torch's GEMMs.  How many of its codes move is
    the reference's own reorder noise -- the yardstick for near-tie differences of any other implementation."""
    perm = np.random.RandomState(seed).permutation(D)
    sdp = dict(sd)
    sdp["centers"] = np.ascontiguousarray(sd["centers"][:, :, perm])
    sdp["to_logits.weight"] = np.ascontiguousarray(sd["to_logits.weight"][:, perm])
    return ref_quantizer(sdp, D, K, N), np.ascontiguousarray(x[:, perm])


def encode_cases(q, sd, x, iters_list, out, reorder=False):
    qp = xp = None
    if reorder:
        D, K, N = int(out["D"]), int(out["K"]), int(out["N"])
        qp, xp = permuted_reference(sd, x, D, K, N)
    for it in iters_list:
        codes = ref_encode(q, x, it, as_bytes=False)
        out[f"codes_it{it}"] = codes.astype(np.uint8 if int(out["K"]) <= 256 else np.uint16)
        if reorder:
            cp = ref_encode(qp, xp, it, as_bytes=False)
            out[f"reorder_noise_it{it}"] = int((cp != codes).any(axis=1).sum())
            print(f"   iters={it}: reference vs its own feature-permuted run: {out[f'reorder_noise_it{it}']} vectors differ")
        c64, margin = search_fp64(sd, x, it)
        out[f"margin_it{it}"] = margin.astype(np.float32)
        nm = int((c64 != codes).any(axis=1).sum())
        print(f"   iters={it}: fp64-vs-fp32 differing vectors {nm}/{len(x)}; margin<2e-6: {(margin < 2e-6).sum()}")
    if int(out["K"]) <= 256:      # (encode(as_bytes=True) asserts codebook_size <= 256, quantization.py:271)
        out["bytes_it%d" % iters_list[-1]] = ref_encode(q, x, iters_list[-1], as_bytes=True)


def decode_cases(q, codes, out):
    with torch.no_grad():
        y = q.decode(torch.from_numpy(codes.astype(np.int64))).numpy()
    out["decode_head"] = y[:16].copy()
    out["decode_rowsum"] = y.astype(np.float64).sum(axis=1)
    out["decode_rowsumsq"] = (y.astype(np.float64) ** 2).sum(axis=1)


def gen_trained(name, D, bytes_per_frame, p1, p2, batch, seed, n_test, x_kind="make_x", iters_list=(0, 1, 2, 5), reorder=False,
                keep=("p1", "p2")):
    """A short CPU training run of the reference trainer; captures the phase-one
    (K=16) and final (K=256) quantizers and their codes on held-out frames."""
    torch.manual_seed(seed)
    random.seed(seed)
    trainer = refq.QuantizerTrainer(dim=D, bytes_per_frame=bytes_per_frame, device=torch.device("cpu"),
                                    phase_one_iters=p1, phase_two_iters=p2)
    it = 0
    phase1 = None
    while not trainer.done():
        if trainer.cur_iter == p1 and phase1 is None:
            phase1 = np_state(trainer.quantizer)   # still K=16: the switch happens at the end of this step
        trainer.step(torch.from_numpy(gen.make_kind(x_kind, 1000 * seed + it, batch, D)))
        it += 1
        if it % 50 == 0:
            print(f"   [{name}] step {it}", flush=True)
    final = np_state(trainer.get_quantizer())
    x = gen.make_kind(x_kind, 777 + seed, n_test, D)
    for tag, sd, K, N in (("p1", phase1, 16, 2 * bytes_per_frame), ("p2", final, 256, bytes_per_frame)):
        if tag not in keep:
            continue
        q = ref_quantizer(sd, D, K, N)
        out = {"D": D, "K": K, "N": N, "x_seed": 777 + seed, "x_kind": x_kind, "B": n_test,
               "x_checksum": gen.checksum(x)}
        for k, v in sd.items():
            out["state." + k] = v
        print(f"[{name}_{tag}] D={D} K={K} N={N}")
        encode_cases(q, sd, x, list(iters_list), out, reorder=reorder)
        decode_cases(q, out["codes_it%d" % iters_list[-1]], out)
        assert iters_list[-1] == 5
        with torch.no_grad():
            yb = q.decode(torch.from_numpy(out["bytes_it5"])).numpy()
            yc = q.decode(torch.from_numpy(out["codes_it5"].astype(np.int64))).numpy()
        assert np.array_equal(yb, yc)
        rel = float(((yc - x) ** 2).sum() / (x ** 2).sum())
        out["rel_err_it5"] = rel
        print(f"   relative reconstruction error {rel:.4f}")
        np.savez_compressed(os.path.join(HERE, f"{name}_{tag}.npz"), **out)


def gen_synth(name, D, K, N, n_test, state_seed, x_seed, iters_list, x_kind="gaussian"):
    sd = gen.synthetic_state(state_seed, D, K, N)
    q = ref_quantizer(sd, D, K, N)
    x = gen.make_gaussian(x_seed, n_test, D) if x_kind == "gaussian" else gen.make_x(x_seed, n_test, D)
    out = {"D": D, "K": K, "N": N, "state_seed": state_seed, "x_seed": x_seed, "x_kind": x_kind, "B": n_test,
           "x_checksum": gen.checksum(x), "centers_checksum": gen.checksum(sd["centers"])}
    # the scaled centers' first row, to pin exp() and the scaling on the test machine
    with torch.no_grad():
        out["scaled_center_0_0"] = q.get_centers()[0, 0].numpy().copy()
    print(f"[{name}] D={D} K={K} N={N} B={n_test}")
    encode_cases(q, sd, x, iters_list, out)
    decode_cases(q, out["codes_it%d" % iters_list[-1]], out)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


def main(which):
    torch.set_num_threads(8)
    if which in ("all", "trained"):
        gen_trained("trained_d64_b4", D=64, bytes_per_frame=4, p1=120, p2=120, batch=256, seed=1, n_test=2048)
        gen_trained("trained_d64_b8", D=64, bytes_per_frame=8, p1=120, p2=120, batch=256, seed=2, n_test=2048)
    if which in ("all", "synth"):
        # every ladder of SURVEY.md section 8a, small dims
        gen_synth("synth_d32_k256_n1", 32, 256, 1, 512, 11, 12, [0, 1, 2])
        gen_synth("synth_d32_k256_n2", 32, 256, 2, 512, 13, 14, [0, 1, 3])
        gen_synth("synth_d64_k256_n16", 64, 256, 16, 512, 15, 16, [0, 1, 3])
        gen_synth("synth_d48_k256_n32", 48, 256, 32, 128, 17, 18, [0, 1, 2])
        gen_synth("synth_d40_k64_n8", 40, 64, 8, 512, 19, 20, [0, 1, 3])      # D not a multiple of 16
        gen_synth("synth_d64_k16_n32", 64, 16, 32, 256, 21, 22, [0, 1, 3])
        gen_synth("synth_d32_k16_n64", 32, 16, 64, 128, 23, 24, [0, 1, 2])
        gen_synth("synth_d30_k32_n4", 30, 32, 4, 512, 25, 26, [0, 1, 3], x_kind="make_x")
    if which in ("all", "wide"):
        # 64 codebooks of more than 16 entries: outside what QuantizerTrainer produces (bytes_per_frame <= 32), inside what
        # the reference's Quantizer accepts; lists of 64 at the two top levels
        gen_synth("synth_d24_k32_n64", 24, 32, 64, 96, 27, 28, [0, 1, 2])
        gen_synth("synth_d16_k256_n64", 16, 256, 64, 48, 29, 30, [0, 1, 2])
    if which in ("all", "k512"):
        # codebooks of 512 and 1,024 entries: Quantizer(codebook_size=...) used with as_bytes=False (quantization.py:35 only
        # asks for a power of two; the trainer never produces them).  One fixture per shape of the combine tree: 1, 2, 4, 8, 16 codebooks
        gen_synth("k512_d24_n1", 24, 512, 1, 256, 41, 42, [0, 1, 2])
        gen_synth("k1024_d24_n2", 24, 1024, 2, 256, 43, 44, [0, 1, 3])
        gen_synth("k512_d32_n4", 32, 512, 4, 256, 45, 46, [0, 1, 3], x_kind="make_x")
        gen_synth("k1024_d40_n8", 40, 1024, 8, 256, 47, 48, [0, 1, 5])
        gen_synth("k512_d16_n16", 16, 512, 16, 128, 49, 50, [0, 1, 2])
    if which in ("all", "stress"):
        # Round 4 ("what's weak" 1): states TRAINED BY THE REFERENCE on frames that are not zero-mean unit-scale
        # Gaussians -- a common offset, one dominant feature, heavy tails -- where the table form's cancellation terms are
        # largest, each with the reference's own reorder noise beside the fp64 margins
        gen_trained("stress_mean10_d64_b8", D=64, bytes_per_frame=8, p1=120, p2=120, batch=256, seed=11, n_test=2048,
                    x_kind="mean10", iters_list=(0, 1, 5), reorder=True)
        gen_trained("stress_mean10_d256_b4", D=256, bytes_per_frame=4, p1=100, p2=100, batch=256, seed=12, n_test=2048,
                    x_kind="mean10", iters_list=(0, 1, 5), reorder=True, keep=("p2",))
        gen_trained("stress_outlier300_d64_b4", D=64, bytes_per_frame=4, p1=120, p2=120, batch=256, seed=13, n_test=2048,
                    x_kind="outlier300", iters_list=(0, 1, 5), reorder=True)
        gen_trained("stress_student2_d64_b4", D=64, bytes_per_frame=4, p1=120, p2=120, batch=256, seed=14, n_test=2048,
                    x_kind="student2", iters_list=(0, 1, 5), reorder=True, keep=("p2",))
        gen_trained("stress_mean100_d64_b4", D=64, bytes_per_frame=4, p1=120, p2=120, batch=256, seed=15, n_test=2048,
                    x_kind="mean100", iters_list=(0, 1, 5), reorder=True, keep=("p2",))
    if which in ("all", "d512"):
        # a state trained by the reference at the BASELINE dim (config B / E shape): 200 + 200 iterations of 256 frames
        gen_trained("trained_d512_b8", D=512, bytes_per_frame=8, p1=200, p2=200, batch=256, seed=16, n_test=2048,
                    x_kind="make_x", iters_list=(0, 1, 5), reorder=True, keep=("p2",))
    if which in ("all", "d512_offset"):
        # Round 5 ("what's weak" 1): the offset case AT THE HEADLINE DIM -- the reference trained on frames with a
        # common offset of 10, dim 512 / 8 bytes, 200 + 200 iterations, 2,048 held-out rows with margins and reorder noise
        gen_trained("stress_mean10_d512_b8", D=512, bytes_per_frame=8, p1=200, p2=200, batch=256, seed=17, n_test=2048,
                    x_kind="mean10", iters_list=(0, 1, 5), reorder=True, keep=("p2",))
    if which in ("all", "configs"):
        # BASELINE.json config shapes (A, B, D) with seeded synthetic states
        gen_synth("config_a_d256_n4", 256, 256, 4, 1024, 101, 102, [0, 1, 5])
        gen_synth("config_b_d512_n8", 512, 256, 8, 4096, 103, 104, [0, 1, 5])
        gen_synth("config_d_d1024_n16", 1024, 256, 16, 512, 105, 106, [0, 1, 5])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
