"""Certification of a row on which an implementation's codes differ from the reference's (SURVEY.md section 7 "hard parts": a
mismatching row must be CERTIFIED as an fp32 near-tie by an fp64 recheck).

What is known per fixture: the reference's indexes after every pass (`refpass`, captured by tests/golden/make_golden_certify.py) and,
from the oracle, the implementation's indexes after every pass (the HIP kernels equal the oracle bit for bit, tested separately).
So the FIRST pass in which the two part is known, and both start it from the same indexes.  certify_row replays that pass in fp64
(fp64_search.replay_pass) twice -- once following the reference's result, once the implementation's -- and reports, node by node of
the combine tree, whether the followed result's ancestor lay inside the fp64 shortlist and, where it did not, by how much
(`rel`: the score gap to the shortlist's boundary, relative to the two competing scores).  A legitimate difference is one where every
such bend is below NEAR: somewhere a candidate sat within fp32 noise of a shortlist's boundary (or of the winner), one side kept it, the
other did not, and whatever follows -- up to a visibly different reconstruction error -- is the search continuing from there.
Pass 0 (the arg max of the logits, :297-301) is certified by the logit gap."""
import numpy as np

from . import fp64_search as f64

NEAR = 1e-6           # a bend below this, relative to the competing scores, is fp32 summation-order noise
OUTCOME = 2e-2        # |SSE(impl) - SSE(ref)| / SSE(ref): what a legitimate boundary flip has been seen to cost or gain (<= 0.7 %)


def _oracle(fx):
    from oracle.oracle import OracleQuantizer
    s = fx["state"]
    return OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"], float(s["logits_scale"]))


def oracle_passes(fx, row, npass):
    """the oracle's indexes of one row after 0 .. npass passes, (npass + 1, N)"""
    o = _oracle(fx)
    x = fx["x"][row:row + 1]
    return np.stack([np.asarray(o.compute_indexes(x, p)).reshape(-1) for p in range(npass + 1)]).astype(np.int64)


def oracle_lists(fx, row, idx_prev):
    """The shortlists the ORACLE keeps in one pass from idx_prev, as tuples: {(level, group): (keep, 2^level) int array}, from its
    trace (mcq_oracle_refine_trace: per prune, the kept candidates' positions in ascending position)."""
    from oracle.oracle import ladder
    o = _oracle(fx)
    N, K = fx["N"], fx["K"]
    t = o.refine_trace(fx["x"][row], idx_prev)
    first, lad = ladder(N, K)
    lists, off = {}, 0
    for n in range(N):
        lists[(0, n)] = t["sel_pos"][off:off + first].astype(np.int64)[:, None]
        off += first
    groups = N
    for v, (kin, kout) in enumerate(lad, start=1):
        groups //= 2
        for g in range(groups):
            pos = t["sel_pos"][off:off + kout].astype(np.int64)
            off += kout
            le, lo = lists[(v - 1, 2 * g)], lists[(v - 1, 2 * g + 1)]
            lists[(v, g)] = np.concatenate([le[pos // kin], lo[pos % kin]], axis=1)
    return lists, np.asarray(t["idx"]).astype(np.int64)


def _lost_at(lists, T, N):
    """first node (level, group), bottom-up, where tuple T's restriction is not in the kept list; None if it survives to the top"""
    v, L = 0, 1
    while True:
        groups = N // L
        for g in range(groups):
            if (v, g) not in lists:
                return None
            if not (lists[(v, g)] == T[g * L:(g + 1) * L][None, :]).all(axis=1).any():
                return (v, g)
        if groups == 1:
            return None
        v, L = v + 1, L * 2


def certify_row(fx, it, row, impl_code=None):
    """Returns a dict describing where and why row `row` differs at `it` passes; raises AssertionError if the difference is not a
    certified near-tie.  impl_code: the implementation's indexes of that row (must be what the oracle gives).

    In the first pass p where the two part (both start it from the same indexes), with F the fp64 result of that pass, each side
    T in (reference R, oracle O) that is not F is explained by decisions within NEAR of a tie, all measured in fp64 relative to
    the two competing scores:
      * T's ancestor lay OUTSIDE an fp64 shortlist and was kept (replay_pass, slack > 0): the bend must be < NEAR;
      * T lay inside every fp64 shortlist but is not the fp64 winner: then its side lost F on the way --
        the oracle's own shortlists (its trace) say at which node: F's ancestor must be within NEAR of the worst candidate the
        oracle kept there; for the reference, whose shortlists are not recorded, F's closest approach to a shortlist boundary
        along the fp64 search must be < NEAR -- or T ties with F at the top-1 within NEAR."""
    assert "refpass" in fx, "fixture without the reference's per-pass indexes (tests/golden/make_golden_certify.py)"
    ref = fx["refpass"][:it + 1, row].astype(np.int64)                 # (it + 1, N)
    imp = oracle_passes(fx, row, it)
    if impl_code is not None:
        assert np.array_equal(np.asarray(impl_code).reshape(-1).astype(np.int64), imp[it]), \
            f"row {row}: the implementation's indexes are not the oracle's -- nothing to certify, that is a bug"
    differ = [p for p in range(it + 1) if not np.array_equal(ref[p], imp[p])]
    assert differ, f"row {row}: no difference after {it} passes"
    p = differ[0]
    C, W, bias, ls = f64.scaled_state(fx["state"])
    N, K, D = C.shape
    x = fx["x"][row]
    out = dict(row=int(row), iters=int(it), first_pass=int(p))
    if p == 0:
        lg = f64.logits_fp64(W, bias, ls, x[None], N, K)[0]            # (N, K)
        n = int(np.flatnonzero(ref[0] != imp[0])[0])
        gap = abs(lg[n, ref[0][n]] - lg[n, imp[0][n]]) / (np.abs(lg[n]).max() + 1e-300)
        out.update(stage="argmax of the logits, codebook %d" % n, gap=float(gap), who="either")
        assert gap < NEAR, out
    else:
        prev = ref[p - 1]
        assert np.array_equal(prev, imp[p - 1])
        F = f64.refine_fp64(C, x[None].astype(np.float64), prev[None])[0][0].astype(np.int64)
        findings = []                        # (gap, description)
        for who, T in (("reference", ref[p]), ("oracle", imp[p])):
            if np.array_equal(T, F):
                continue
            nodes = f64.replay_pass(C, x, prev, T)
            bends = [nd for nd in nodes if nd["slack"] > 0 and nd["keep"] > 1]
            for nd in bends:
                findings.append((nd["rel"], "pass %d, level %d, group %d (prune to %d of %d): the %s kept a candidate of fp64 rank %d"
                                 % (p, nd["level"], nd["group"], nd["keep"], nd["n_candidates"], who, nd["rank"])))
            top = nodes[-1]
            if top["rank"] > 0 and not bends:
                # inside every fp64 shortlist, not the fp64 winner: a tie at the top-1, or this side lost F below
                if top["rel"] < NEAR:
                    findings.append((top["rel"], "pass %d, top-1: the %s's winner ties with the fp64 winner" % (p, who)))
                elif who == "oracle":
                    lists, o_idx = oracle_lists(fx, row, prev)
                    assert np.array_equal(o_idx, T)
                    node = _lost_at(lists, F, N)
                    assert node is not None, (out, "the oracle kept the fp64 winner to the top and chose a clearly worse one")
                    v, g = node
                    fF = f64.tuple_score(C, x, prev, v, g, F[g << v:(g + 1) << v])
                    worst_kept = max(f64.tuple_score(C, x, prev, v, g, t) for t in lists[(v, g)])
                    rel = (worst_kept - fF) / (max(abs(worst_kept), abs(fF)) + 1e-300)
                    # (rel > 0: the oracle kept a candidate that fp64 ranks below F's ancestor, by that much)
                    findings.append((abs(rel), "pass %d, level %d, group %d (prune to %d): the oracle dropped the fp64 winner's ancestor for a "
                                     "candidate %.2e worse" % (p, v, g, len(lists[(v, g)]), rel)))
                    assert rel >= 0, (out, findings, "the oracle dropped a candidate that fp64 ranks inside its shortlist by a clear margin")
                else:
                    nodesF = f64.replay_pass(C, x, prev, F)
                    # F's closest approach to a boundary: slack is f(F|g) - f(last kept); the first dropped one is not recorded,
                    # so bound it by the gap of the pass's stored margin for this row (margin2: smallest boundary gap of the search)
                    m2 = float(fx[f"margin2_it{it}"][row]) if f"margin2_it{it}" in fx else None
                    findings.append((m2 if m2 is not None else 1.0,
                                     "pass %d: the reference lost the fp64 winner at a shortlist boundary (closest boundary gap of the row)" % p))
                    del nodesF
        assert findings, (out, "both results are the fp64 result, yet they differ")
        gap, stage = max(findings)
        out.update(stage=stage, gap=float(gap), who=stage, findings=findings)
        assert gap < NEAR, out
    # outcome: the reconstruction errors of the two final codes
    sse_ref = float(fx[f"sse64_it{it}"][row]) if f"sse64_it{it}" in fx else float(f64.sse_fp64(fx["state"], x[None], ref[it][None])[0])
    sse_imp = float(f64.sse_fp64(fx["state"], x[None], imp[it][None])[0])
    out["sse_rel"] = (sse_imp - sse_ref) / sse_ref
    assert abs(out["sse_rel"]) <= OUTCOME, out
    return out
