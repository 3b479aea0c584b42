"""The index search of Quantizer._compute_indexes (/root/reference/quantization/quantization.py:281-547) carried out in fp64
with numpy -- no import of the reference, so it runs wherever the tests run.  Two users:

  * tests/golden/make_golden.py / make_golden_certify.py (build container): the per-row decision margins stored in the fixtures;
  * tests/golden/certify.py: the replay that certifies a row on which two implementations' codes differ.

Scores: with o_m the current entries and x_err = sum_m o_m - x (:340), the score of replacing the entries of a group g of
codebooks by a tuple t is  f_g(t) = |x_err + sum_{n in g} (C[n, t_n] - o_n)|^2  -- :418 for single codebooks, :533-535 for
pairs of groups (the identity |a + b|^2 = |a|^2 + |b|^2 - |x_err|^2 + 2 (a - x_err).(b - x_err) with a = x_err + delta_e,
b = x_err + delta_o).  Every sort-and-truncate (:470-503) keeps the K_cutoff smallest f of its candidates."""
import numpy as np


def k_cutoff(K, L):
    kc = 8 if K <= 16 else 16
    while L >= 4:
        L //= 4
        kc *= 2
    return min(kc, 128)


def scale_factors(sd):
    """(exp(10 centers_scale), exp(10 logits_scale)) as fp32: the factors pinned in the state (`scales_exp`, captured from the
    reference's own torch on the machine that generated the fixture) when it carries them -- torch's fp32 exp differs in the last
    bit between CPUs (measured: Xeon ...47, EPYC ...46 for the same argument), and one bit of the scale moves near-tie codes --
    else torch's exp on this host, else numpy's."""
    pinned = getattr(sd, "scales_exp", None)      # (fixtures.PinnedState)
    if pinned is not None:
        return np.float32(pinned[0]), np.float32(pinned[1])
    try:
        import torch
        cs = np.float32((torch.tensor(float(sd["centers_scale"]), dtype=torch.float32) * 10.0).exp().item())
        ls = np.float32((torch.tensor(float(sd["logits_scale"]), dtype=torch.float32) * 10.0).exp().item())
    except ImportError:
        cs = np.exp(np.float32(sd["centers_scale"]) * np.float32(10.0)).astype(np.float32)
        ls = np.exp(np.float32(sd["logits_scale"]) * np.float32(10.0)).astype(np.float32)
    return cs, ls


def scaled_state(sd):
    """(C (N,K,D) fp64 of the fp32 scaled centers, W fp64, bias fp64, logits scale fp32) as get_centers() / _logits() form them
    (:77-79, :277-279): the scale factors are fp32 exp()s and the scaled centers fp32 products."""
    cs, ls = scale_factors(sd)
    C = (cs * sd["centers"].astype(np.float32)).astype(np.float32).astype(np.float64)
    return C, sd["to_logits.weight"].astype(np.float64), sd["to_logits.bias"].astype(np.float64), ls


def logits_fp64(W, bias, ls, x32, N, K):
    sx = (ls * x32.astype(np.float32)).astype(np.float32).astype(np.float64)
    return (sx @ W.T + bias).reshape(-1, N, K)


def _gap(sorted_vals, keep, norm):
    """gap between the last kept and the first dropped candidate, relative to `norm` (the round-1 normalisation: |x|^2 + E)"""
    if sorted_vals.shape[-1] <= keep:
        return np.full(sorted_vals.shape[:-1], np.inf)
    return (sorted_vals[..., keep] - sorted_vals[..., keep - 1]) / norm


def _gap2(sorted_vals, keep):
    """the same gap relative to the two competing scores themselves (round 6): what an fp32 evaluation of THESE scores can resolve"""
    if sorted_vals.shape[-1] <= keep:
        return np.full(sorted_vals.shape[:-1], np.inf)
    a, b = sorted_vals[..., keep - 1], sorted_vals[..., keep]
    return (b - a) / (np.maximum(np.abs(a), np.abs(b)) + 1e-300)


def search_fp64(sd, x, iters, per_pass=False):
    """returns (codes (B,N) int64, margin (B,), margin2 (B,)) -- margin: smallest decision gap along the fp64 search relative to
    |x|^2 + E (as stored since round 1), margin2: relative to the competing scores (round 6).  per_pass: also the codes after
    every pass, (iters + 1, B, N)."""
    C, W, bias, ls = scaled_state(sd)
    N, K, D = C.shape
    B = x.shape[0]
    codes = np.zeros((B, N), np.int64)
    margin, margin2 = np.zeros(B), np.zeros(B)
    passes = np.zeros((iters + 1, B, N), np.int64)
    step = max(1, min(B, (1 << 27) // (N * 32 * D * 8)))
    for lo in range(0, B, step):
        xb32 = x[lo:lo + step]
        xb = xb32.astype(np.float64)
        logits = logits_fp64(W, bias, ls, xb32, N, K)
        srt = np.sort(logits, axis=2)
        m = ((srt[..., -1] - srt[..., -2]) / (np.abs(srt).max(axis=2) + 1e-300)).min(axis=1)
        m2 = m.copy()
        idx = logits.argmax(axis=2)
        passes[0, lo:lo + step] = idx
        for p in range(iters):
            idx, mi, mi2 = refine_fp64(C, xb, idx)
            m, m2 = np.minimum(m, mi), np.minimum(m2, mi2)
            passes[p + 1, lo:lo + step] = idx
        codes[lo:lo + step] = idx
        margin[lo:lo + step] = m
        margin2[lo:lo + step] = m2
    if per_pass:
        return codes, margin, margin2, passes
    return codes, margin, margin2


def refine_fp64(C, x, idx):
    """one _refine_indexes pass (:308-547) for a batch, in fp64: (new idx, margin, margin2)"""
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
    Ng, L = N, 1
    keep = 1 if Ng == 1 else k_cutoff(K, L)
    order = np.argsort(S, axis=2, kind="stable")
    Ss = np.take_along_axis(S, order, axis=2)
    margin = _gap(Ss, keep, norm[:, None]).min(axis=1)
    margin2 = _gap2(Ss, keep).min(axis=1)
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
        margin2 = np.minimum(margin2, _gap2(Ss, keep).min(axis=1))
        sel = order[..., :keep]
        curS = Ss[..., :keep]
        a, b = sel // Kg, sel % Kg
        te = np.take_along_axis(tuples[:, 0::2], a[..., None], axis=2)
        to = np.take_along_axis(tuples[:, 1::2], b[..., None], axis=2)
        tuples = np.concatenate([te, to], axis=3)
        deltas = (np.take_along_axis(de, a[..., None], axis=2) + np.take_along_axis(do, b[..., None], axis=2))
        Ng, Kg = newN, keep
    return tuples[:, 0, 0, :], margin, margin2


def sse_fp64(sd, x, codes):
    """|sum_n C[n, code_n] - x|^2 per row, fp64: the quantity the search minimises (its outcome)"""
    C = scaled_state(sd)[0]
    N = C.shape[0]
    rec = C[np.arange(N)[None, :], codes.astype(np.int64)].sum(axis=1)
    return ((rec - x.astype(np.float64)) ** 2).sum(axis=1)


# ---------------------------------------------------------------------------------------------------------------------
# one vector, one pass, FOLLOWING a given result: where does the fp64 search have to bend to reach it, and by how much?
# ---------------------------------------------------------------------------------------------------------------------
def replay_pass(C, x, idx_prev, follow):
    """The fp64 pass from idx_prev for ONE vector, with the tuple `follow` (N,) kept alive at every sort-and-truncate: at each node
    (level v, group g) the candidates are the products of the children's kept lists (the fp64 top-keep, plus the followed tuple's
    restriction if it fell outside).  Returns a list of dicts, one per node in the order the search visits them:
        level, group, keep, n_candidates, rank (of the followed tuple among the candidates, 0 = best),
        slack = f(followed) - f(last kept candidate of the fp64 order)   (<= 0: fp64 keeps it anyway),
        rel   = slack / max(|f(followed)|, |f(boundary)|),
        f_follow, f_boundary.
    The largest positive `rel` is how far from the fp64-optimal decision an implementation had to be, somewhere, to end at `follow`."""
    N, K, D = C.shape
    x = x.astype(np.float64)
    old = C[np.arange(N), idx_prev]                 # (N,D)
    xerr = old.sum(axis=0) - x
    E = float(xerr @ xerr)
    nodes = []
    # level-0 candidates: all K entries of each codebook
    lists = []                                     # per group: (tuples (cnt, L), deltas (cnt, D), f (cnt,))
    L = 1
    keep = 1 if N == 1 else k_cutoff(K, L)
    for n in range(N):
        delta = C[n] - old[n][None, :]             # (K,D)
        y = xerr[None, :] + delta
        f = (y * y).sum(-1)
        lists.append(_truncate(nodes, 0, n, keep, np.arange(K)[:, None], delta, f, np.asarray([follow[n]])))
    Ng = N
    v = 0
    while Ng > 1:
        v += 1
        L *= 2
        newN = Ng // 2
        keep = 1 if newN == 1 else k_cutoff(K, L)
        nxt = []
        for g in range(newN):
            te, de, fe = lists[2 * g]
            to, do, fo = lists[2 * g + 1]
            tup = np.concatenate([np.repeat(te, len(to), axis=0), np.tile(to, (len(te), 1))], axis=1)
            delta = (de[:, None, :] + do[None, :, :]).reshape(-1, D)
            f = ((fe[:, None] + fo[None, :]) - E + 2.0 * (de @ do.T)).reshape(-1)
            nxt.append(_truncate(nodes, v, g, keep, tup, delta, f, follow[g * L:(g + 1) * L]))
        lists = nxt
        Ng = newN
    return nodes


def _truncate(nodes, level, group, keep, tup, delta, f, follow_g):
    order = np.argsort(f, kind="stable")
    hit = np.flatnonzero((tup == follow_g[None, :]).all(axis=1))
    assert hit.size >= 1, "the followed tuple is not among the candidates (its halves were kept alive below)"
    fi = int(hit[0])
    rank = int(np.flatnonzero(order == fi)[0])
    kb = min(keep, len(f)) - 1
    f_follow, f_bound = float(f[fi]), float(f[order[kb]])
    slack = f_follow - f_bound
    nodes.append(dict(level=level, group=group, keep=keep, n_candidates=len(f), rank=rank, slack=slack,
                      rel=slack / (max(abs(f_follow), abs(f_bound)) + 1e-300), f_follow=f_follow, f_boundary=f_bound))
    kept = list(order[:keep])
    if rank >= keep:
        kept.append(fi)                            # bend: keep the followed tuple alive
    kept = np.asarray(kept)
    return tup[kept], delta[kept], f[kept]


def tuple_score(C, x, idx_prev, level, group, tup):
    """f_g(t) of one tuple of a level-`level` group (module docstring), fp64"""
    N = C.shape[0]
    old = C[np.arange(N), idx_prev]
    xerr = old.sum(axis=0) - x.astype(np.float64)
    L = 1 << level
    y = xerr.copy()
    for j, n in enumerate(range(group * L, (group + 1) * L)):
        y += C[n, int(tup[j])] - old[n]
    return float(y @ y)


# ---------------------------------------------------------------------------------------------------------------------
# one vector, one pass, with the search's NEAR-TIES as choice points: is a given result an outcome of the search under some
# resolution of them?
# ---------------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    return abs(a - b) / (max(abs(a), abs(b)) + 1e-300)


def _truncate_choice(level, group, keep, tup, delta, f, near, choices, ties):
    """keep `keep` of the candidates: the fp64 order, except that candidates within `near` (relative) of the boundary -- of the
    best one when keep == 1 -- may be exchanged: choices[(level, group)] picks which (0 = the fp64 order)."""
    order = np.argsort(f, kind="stable")
    n = len(f)
    if n <= keep:
        return tup[order], delta[order], f[order]
    # the cluster of ranks round the boundary keep - 1 | keep whose members are within `near` of a boundary score
    lo, hi = keep - 1, keep
    if _rel(f[order[lo]], f[order[hi]]) >= near:
        return tup[order[:keep]], delta[order[:keep]], f[order[:keep]]
    while lo > 0 and _rel(f[order[lo - 1]], f[order[keep]]) < near and keep - lo < 3:
        lo -= 1
    while hi + 1 < n and _rel(f[order[hi + 1]], f[order[keep - 1]]) < near and hi - keep < 2:
        hi += 1
    import itertools
    members = list(range(lo, hi + 1))
    need = keep - lo
    combos = list(itertools.combinations(members, need))      # combos[0] = the fp64 order's own choice
    pick = choices.get((level, group), 0)
    gap = _rel(f[order[keep - 1]], f[order[keep]])
    ties.append(((level, group), len(combos), gap))
    ranks = list(range(lo)) + list(combos[min(pick, len(combos) - 1)])
    sel = order[np.asarray(ranks, dtype=np.int64)]
    return tup[sel], delta[sel], f[sel]


def pass_with_choices(C, x, idx_prev, choices, near):
    """the fp64 pass of ONE vector from idx_prev with the near-ties at the nodes in `choices` resolved as given;
    returns (result (N,), [(node, alternatives, gap), ...] of every near-tie met on the way)"""
    N, K, D = C.shape
    x = x.astype(np.float64)
    old = C[np.arange(N), idx_prev]
    xerr = old.sum(axis=0) - x
    E = float(xerr @ xerr)
    ties = []
    L = 1
    keep = 1 if N == 1 else k_cutoff(K, L)
    lists = []
    for n in range(N):
        delta = C[n] - old[n][None, :]
        y = xerr[None, :] + delta
        lists.append(_truncate_choice(0, n, keep, np.arange(K)[:, None], delta, (y * y).sum(-1), near, choices, ties))
    Ng, v = N, 0
    while Ng > 1:
        v += 1
        L *= 2
        newN = Ng // 2
        keep = 1 if newN == 1 else k_cutoff(K, L)
        nxt = []
        for g in range(newN):
            te, de, fe = lists[2 * g]
            to, do, fo = lists[2 * g + 1]
            tup = np.concatenate([np.repeat(te, len(to), axis=0), np.tile(to, (len(te), 1))], axis=1)
            delta = (de[:, None, :] + do[None, :, :]).reshape(-1, D)
            f = ((fe[:, None] + fo[None, :]) - E + 2.0 * (de @ do.T)).reshape(-1)
            nxt.append(_truncate_choice(v, g, keep, tup, delta, f, near, choices, ties))
        lists = nxt
        Ng = newN
    return lists[0][0][0].astype(np.int64), ties


def explain_by_near_ties(C, x, idx_prev, target, near, budget=400, max_flips=4):
    """Is `target` what the fp64 pass from idx_prev gives under SOME resolution of its near-ties (decisions whose competing scores are
    within `near`, relative)?  Breadth-first over sets of flipped near-ties, smallest sets first.  Returns the list of flips
    [((level, group), gap), ...] (empty: target IS the fp64 result) or None."""
    target = np.asarray(target).astype(np.int64)
    frontier = [dict()]
    seen = set()
    runs = 0
    for depth in range(max_flips + 1):
        nxt = []
        for ch in frontier:
            key = tuple(sorted(ch.items()))
            if key in seen:
                continue
            seen.add(key)
            res, ties = pass_with_choices(C, x, idx_prev, ch, near)
            runs += 1
            gaps = dict((node, gap) for node, _, gap in ties)
            if np.array_equal(res, target):
                return [(node, gaps.get(node, 0.0)) for node in sorted(ch)]
            if runs >= budget:
                return None
            for node, alts, _ in ties:
                if node in ch:
                    continue
                for a in range(1, alts):
                    c2 = dict(ch)
                    c2[node] = a
                    nxt.append(c2)
        frontier = nxt
        if not frontier:
            break
    return None
