"""Certification of a row on which an implementation's codes differ from the reference's (SURVEY.md section 7 "hard parts": a
mismatching row must be CERTIFIED as an fp32 near-tie by an fp64 recheck).

What is known per fixture: the reference's indexes after every pass (`refpass`, captured by tests/golden/make_golden_certify.py), the
fp32 scale factors it computed with (`scales_exp`) and, from the oracle, the implementation's indexes after every pass (the HIP kernels
equal the oracle bit for bit, tested separately).  So the FIRST pass in which the two part is known, and both start it from the same
indexes.  certify_row runs that pass in fp64 with the search's NEAR-TIES as choice points (fp64_search.explain_by_near_ties): a decision
-- a sort-and-truncate's boundary, the final top-1 -- whose two competing scores lie within NEAR of each other, relatively, may go
either way.  A legitimate difference is one where EACH side's result is what the fp64 search returns under some resolution of its
near-ties; the certificate names those decisions (level and group of the combine tree) and their gaps.  Whatever follows a flipped
near-tie -- other candidates in later shortlists, up to a visibly different reconstruction error -- is the search continuing from there,
which is why the end results are compared by outcome only loosely (OUTCOME).
Pass 0 (the arg max of the logits, :297-301) is certified by the logit gap."""
import numpy as np

from . import fp64_search as f64

NEAR = 1e-6           # a bend below this, relative to the competing scores, is fp32 summation-order noise
OUTCOME = 5e-2        # |SSE(impl) - SSE(ref)| / SSE(ref): what a flipped near-tie has been seen to cost or gain downstream (<= 1.5 %)


def _oracle(fx):
    from oracle.oracle import OracleQuantizer
    s = fx["state"]
    return OracleQuantizer(s["centers"], float(s["centers_scale"]), s["to_logits.weight"], s["to_logits.bias"], float(s["logits_scale"]),
                           scales_exp=getattr(s, "scales_exp", None))


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


def certify_row(fx, it, row, impl_code=None):
    """Returns a dict describing where and why row `row` differs at `it` passes; raises AssertionError if the difference is not a
    certified near-tie.  impl_code: the implementation's indexes of that row (must be what the oracle gives)."""
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
        findings = []
        for who, T in (("reference", ref[p]), ("oracle", imp[p])):
            flips = f64.explain_by_near_ties(C, x, prev, T, NEAR)
            assert flips is not None, (out, f"the {who}'s result of pass {p} is not an outcome of the fp64 search under any resolution "
                                            f"of its near-ties (gaps < {NEAR:g})")
            out["flips_" + who] = [(lv, g, float(gap)) for (lv, g), gap in flips]
            for (lv, g), gap in flips:
                findings.append((float(gap), "pass %d, level %d, group %d: the %s resolves this near-tie against the fp64 order" % (p, lv, g, who)))
        assert findings, (out, "both results are the fp64 result, yet they differ")
        gap, stage = max(findings)
        out.update(stage=stage, gap=float(gap), findings=findings)
        assert gap < NEAR, out
    # outcome: the reconstruction errors of the two final codes
    sse_ref = float(fx[f"sse64_it{it}"][row]) if f"sse64_it{it}" in fx else float(f64.sse_fp64(fx["state"], x[None], ref[it][None])[0])
    sse_imp = float(f64.sse_fp64(fx["state"], x[None], imp[it][None])[0])
    out["sse_rel"] = (sse_imp - sse_ref) / sse_ref
    assert abs(out["sse_rel"]) <= OUTCOME, out
    return out
