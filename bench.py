#!/usr/bin/env python
"""Throughput of the encode hot path on MI355X: vectors encoded per second at
dim=512, 8 codebooks of 256 entries, 5 refinement passes (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...        (no launcher: bench.py starts its N ranks itself through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one Quantizer.encode() of the per-GPU batch (65,536 synthetic Gaussian
vectors already resident in HBM; uint8 codes written to HBM).  With N GPUs every rank
encodes its own shard -- no collective on the data path (weak scaling).  Rank 0 prints
ONE JSON line.  `roofline` is for the dominant kernel: its launch time from HIP events on the
launch stream (mcq_profile_encode), its counters (L1 <- L2 requests, HBM-side bytes) read live by
rocprofv3 --pmc child runs of this command (--no-pmc-check: from the committed passes); a table
kernel is priced against what binds it, the L2 -> L1 path (`bound`: "l2"); `cpu_baseline` times the torch-CPU restatement of the
reference's op sequence (oracle/torch_port.py) on this host's cores on a bounded sample.
Secondary objects: `parity` (sampled rows vs the oracle, the 4,096 reference-fixture rows),
`configs` (BASELINE configs A, D and config C's per-GPU shard), `decode`, `trainer_step`,
and -- under WORLD_SIZE > 1 -- `dp_trainer` (QuantizerTrainer.step with the RCCL all-reduce).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_I8_MFMA_TOPS = 5033.0     # 2 x the dense bf16 peak (MI355X_MICROARCH.md: i8 = 2x bf16 rate; 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz)
LIMB_PRODUCTS = 10             # i8 MFMA products per fixed-point product step (mcq_fix_kernels.h)
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec
NEAR_TIE = 2e-6                # tests/golden/fixtures.py


def k_cutoff(K, L):
    kc = 8 if K <= 16 else 16
    while L >= 4:
        L //= 4
        kc *= 2
    return min(kc, 128)


def reference_flops_per_vector(D, N, K, iters):
    """matmul FLOPs (2*m*n*k) of the REFERENCE algorithm per vector, SURVEY.md 8(d): logits, and per pass the
    stage-0 GEMM plus every pair stage."""
    per_pass = 2.0 * D * N * K
    G, L, KI = N, 1, (1 if N == 1 else k_cutoff(K, 1))
    while G > 1:
        Gout = G // 2
        per_pass += Gout * KI * KI * 2.0 * D
        KI = 1 if Gout == 1 else k_cutoff(K, 2 * L)
        G, L = Gout, 2 * L
    return 2.0 * D * N * K + iters * per_pass


PEAK_L2_GBPS = 34500.0         # MI355X_MICROARCH.md, L2 (per XCD 4 MiB): ~34.5 TB/s aggregate L2 -> L1


def kernel_work(B, D, N, K):
    """Per launch of each category mcq_profile_encode reports (keys = mcq_profile_category_name): multiply-adds x 2 of the
    product it forms, algorithmic HBM bytes, table bytes.  The two products (logits, x.C) are the only matrix-core work: exact
    fixed-point products, ten i8 MFMA limb products per multiply-add.  A table kernel's HBM bytes are what it must exchange
    with memory (per-vector inputs, lists and tables written and read back); its table bytes are the 4-byte Gram entries /
    Gram row segments it reads, which an XCD's L2 serves (the Gram matrix is resident state: 16 MB at 8 x 256)."""
    gemm = 2.0 * D * N * K * B
    kc = [k_cutoff(K, 1 << v) for v in range(6)]
    leaf = (kc[0] * kc[0] + 2 * kc[0] + 1) * 4.0           # one full leaf table's Gram reads (x 0.3 below: the lazy
    # level-1 tables read ~ 85 of the 289 entries)
    lists0 = 2 * kc[0] * 5.0                               # two level-0 lists read (entry + score)
    dq = (D + 127) // 128 * 128
    pair1 = (B * (N / 4) * (2 * lists0 + 2 * kc[1] * 6.0 + kc[2] * 6.0), B * (N / 4) * 4 * leaf * 0.3)
    tab1 = (B * 4 * (N / 8) * (2 * lists0 + 2 * kc[1] * 6.0 + kc[1] * kc[1] * 4.0), B * 4 * (N / 8) * 4 * leaf * 0.3)
    return {
        "logits_product_argmax": (gemm, B * (dq * 4.0 + N), 0.0),
        "frames_to_limbs": (0.0, B * (D * 4.0 + 2 * dq * 4.0 + 12), 0.0),        # frames read, two sets of limb planes written
        "stage0_tables": (0.0, B * N * (K * 4.0 + kc[0] * 5.0 + 5), B * N * N * K * 4.0),
        "xc_product": (gemm, B * (dq * 4.0 + N * K * 4.0), 0.0),
        "combine_level0": (0.0, B * (N / 2) * (lists0 + kc[1] * 6.0), B * (N / 2) * leaf),
        "combine_level1": (0.0,) + pair1,
        "tables_level1": (0.0,) + tab1,
        "level1_combines_and_tables": (0.0, pair1[0] + tab1[0], pair1[1] + tab1[1]),
        "combine_level2": (0.0, B * (N / 8) * (4 * kc[1] * kc[1] * 4.0 + 2 * kc[2] * 6.0 + kc[3] * 6.0) + B * (N + N * 4.0 + 8), B * (N * N + N) * 4.0),
        "tables_upper_levels": (0.0, B * max(N // 16, 0) * 16 * kc[1] * kc[1] * 4.0, B * max(N // 16, 0) * 16 * 4 * leaf * 0.3),
        "combine_upper_levels": (0.0, B * max(N // 16, 0) * 16 * kc[1] * kc[1] * 4.0, 0.0),
        "residual_energies": (0.0, B * (N + N * 4.0 + 8), B * (N * N + N) * 4.0),       # E, R from the tables
        "encode_tail": (0.0, B * N * 2.0, 0.0),
    }


# kernel-name prefixes of the launch categories (rocprofv3's Kernel_Name without `void mcq::`), for the --pmc passes
CATEGORY_KERNELS = {"logits_product_argmax": "k_fgemm<1>", "xc_product": "k_fgemm<0>", "frames_to_limbs": "k_fix_rows<",
                    "residual_energies": "k_tf_er<", "stage0_tables": "k_tf_stage0", "combine_level0": "k_tf_pair0",
                    "combine_level1": "k_tf_pair1<", "tables_level1": "k_tf_table1<", "combine_level2": "k_tf_comb<",
                    "level1_combines_and_tables": "k_tf_level1<", "tables_upper_levels": "k_tf_up<",
                    "combine_upper_levels": "k_tf_comb3<"}


def category_kernels(N):
    """the map for a run whose every launch has N codebooks: with 16 the cousin tables of level 3 are a k_tf_table1 launch of
    their own (category tables_upper_levels); the level-2 ones ride in k_tf_level1"""
    m = dict(CATEGORY_KERNELS)
    if N == 16:
        m["tables_upper_levels"] = "k_tf_table1<"
        del m["tables_level1"]
    return m


def profile_kernels(L, q, x, B, D, N, K, iters, dev, reps=3):
    """{category: launches, avg ms, work figures} of one encode: mcq_profile_encode (HIP events on the launch stream)."""
    with torch.no_grad():          # inference flavour of the derived state (host-side scale factors)
        blob = q._prepared()
    ws = q._workspace(B, dev)
    CAP = 32
    ms = (ctypes.c_float * CAP)()
    cnt = (ctypes.c_int * CAP)()
    acc, launches = np.zeros(CAP), np.zeros(CAP, np.int64)
    st = torch.cuda.current_stream(dev).cuda_stream
    ncat = 0
    for _ in range(reps):
        ncat = L.mcq_profile_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, iters, ws.data_ptr(),
                                    ws.numel(), st, ms, cnt, CAP)
        assert ncat > 0, ncat
        acc[:ncat] += np.array(ms[:ncat])
        launches[:ncat] = np.array(cnt[:ncat])
    acc /= reps
    # a large per-vector workspace share cuts the batch into chunks (dim 1024 / 16 codebooks: two of 32,768): every launch
    # then covers one chunk, and so must the work figures it is priced with
    names = [L.mcq_profile_category_name(i).decode() for i in range(ncat)]
    chunks = max(1, int(launches[names.index("frames_to_limbs")])) if "frames_to_limbs" in names else 1
    work = kernel_work(B // chunks, D, N, K)
    kernels = {}
    for i in range(ncat):
        if launches[i] == 0:
            continue
        name = names[i]
        fl, by, tb = work[name]
        avg_ms = max(acc[i], 1e-9) / launches[i]
        kernels[name] = {"launches_per_encode": int(launches[i]), "avg_ms": round(float(avg_ms), 4),
                         "ms_per_encode": round(float(acc[i]), 3)}
        if fl > 0:
            tf_ = fl / (avg_ms * 1e-3) / 1e12
            kernels[name].update(gflop_per_launch=round(fl / 1e9, 2), f32_equivalent_tflops=round(tf_, 2),
                                 i8_tops=round(LIMB_PRODUCTS * tf_, 1), frac_of_i8_mfma_peak=round(LIMB_PRODUCTS * tf_ / PEAK_I8_MFMA_TOPS, 4),
                                 hbm_gbyte_per_launch=round(by / 1e9, 3))
        else:
            kernels[name].update(hbm_gbyte_per_launch=round(by / 1e9, 3), hbm_gbytes_per_s=round(by / (avg_ms * 1e-3) / 1e9, 1))
            if tb > 0:
                kernels[name].update(table_gbyte_per_launch=round(tb / 1e9, 3), table_gbytes_per_s_from_l2=round(tb / (avg_ms * 1e-3) / 1e9, 1))
        kernels[name]["vectors_per_launch"] = B // chunks
    return kernels


def pmc_live(D, N, B, iters, budget_s=200.0):
    """Counters of THIS command's kernels, read by rocprofv3 child runs of it (kernel-trace + one counter group per run:
    TCP_TCC_READ_REQ / FETCH_SIZE / WRITE_SIZE; no other trace domain).  {category: {...}} or None when rocprofv3 is not
    there or the budget ran out before the first pass.  Bytes as MI355X_MICROARCH.md corrects them for gfx950:
    HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB, one L1 <- L2 request = one 128-byte line."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out_root = tempfile.mkdtemp(prefix="mcq_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    t0 = time.time()
    done = []
    for tag, counters in (("tcp", ["TCP_TCC_READ_REQ_sum"]), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        left = budget_s - (time.time() - t0)
        if left < 30:
            break
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", os.path.join(out_root, tag), "--",
               sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-profile",
               "--no-secondary", "--no-pmc-check", "--dim", str(D), "--num-codebooks", str(N), "--batch-per-gpu", str(B),
               "--refine-iters", str(iters)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
        except (subprocess.TimeoutExpired, OSError):
            break
        if r.returncode != 0:
            continue
        for f in glob.glob(os.path.join(out_root, tag, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("mcq::", "")
                vals[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
        done.append(tag)
    shutil.rmtree(out_root, ignore_errors=True)
    if not done:
        return None
    out = {}
    for cat, prefix in category_kernels(N).items():
        names = [n for n in vals if n.startswith(prefix)]
        if not names:
            continue
        def mean(counter):
            v = [x_ for n_ in names for x_ in vals[n_][counter]]
            return (sum(v) / len(v), len(v)) if v else (None, 0)
        rq, nl = mean("TCP_TCC_READ_REQ_sum")
        fk, _ = mean("FETCH_SIZE")
        wk, _ = mean("WRITE_SIZE")
        out[cat] = {"kernel": names[0], "launches_averaged": nl,
                    "traffic_bytes": None if fk is None or wk is None else int(round((2 * fk + wk) * 1024)),
                    "l2_read_requests": None if rq is None else int(round(rq)),
                    "l2_read_request_bytes": None if rq is None else int(round(rq * 128))}
    out["_passes"] = done
    out["_seconds"] = round(time.time() - t0, 1)
    return out


def pmc_committed(dom_name, D, N, K, B, iters):
    """the newest committed --pmc passes of EXACTLY this workload (profiles/rNN_pmc_traffic*.json carry their shape as `_shape`;
    files from before round 6 are the headline shape when they have no suffix and (dim, codebooks, 256, 65,536, 5) of their
    `_d<dim>_n<codebooks>` suffix otherwise), when the live read is off.  Counters of another batch size or pass count would
    price this launch time against the wrong bytes: no match, no counters."""
    import glob
    import re
    want = {"D": D, "N": N, "K": K, "B": B, "iters": iters}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_pmc_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        shape = pmc.get("_shape")
        if shape is None:
            m = re.search(r"_pmc_traffic(?:_d(\d+)_n(\d+))?\.json$", f)
            shape = {"D": int(m.group(1)) if m.group(1) else 512, "N": int(m.group(2)) if m.group(2) else 8, "K": 256, "B": 65536, "iters": 5}
        if shape == want and dom_name in pmc:
            return pmc[dom_name], "committed: profiles/%s (separate --pmc runs of this command; not counters of this run)" % os.path.basename(f)
    return None, None


def roofline_of(dom_name, kern, work, pmc, pmc_src):
    """`roofline` of the dominant launch category, ONE definition for every round: `frac` = the kernel's USEFUL work per launch /
    its launch time / the peak of the resource named by `bound`.
      * a product (`bound: "mfma"`): i8 MFMA operations against the dense i8 peak;
      * a table kernel (`bound: "hbm"`): its algorithmic HBM bytes (per-vector inputs, lists and tables written and read back:
        SURVEY 8d's per-vector figure x the vectors of a launch) against the HBM peak.  It has no FLOPs.
    What actually limits a table kernel is reported BESIDE that, never as `frac`: `l2.useful_frac` (the 4-byte Gram entries it
    uses against the L2 -> L1 peak) and `l2.line_saturation` (the 128-byte lines its gathers move, TCP_TCC_READ_REQ x 128 B,
    against the same peak: a gauge of how full that path is, most of it waste)."""
    dom_fl, dom_by, dom_tb = work
    dom_ms = kern["avg_ms"]
    traffic = pmc.get("traffic_bytes") if pmc else None
    lines = pmc.get("l2_read_request_bytes") if pmc else None
    if dom_fl > 0:
        achieved = LIMB_PRODUCTS * dom_fl / (dom_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 1), "peak": PEAK_I8_MFMA_TOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_I8_MFMA_TOPS, 4), "traffic": traffic,
                "traffic_source": pmc_src, "gop_per_launch": round(LIMB_PRODUCTS * dom_fl / 1e9, 2),
                "avg_launch_ms": round(float(dom_ms), 4),
                "note": "i8 MFMA operations (ten limb products per multiply-add of the exact fixed-point product) against "
                        "the dense i8 peak; unit reads TOP/s"}
    hbm_ach = dom_by / (dom_ms * 1e-3) / 1e9
    useful = dom_tb / (dom_ms * 1e-3) / 1e9
    l2 = {"peak": PEAK_L2_GBPS, "unit": "GB/s", "useful_table_gbyte_per_launch": round(dom_tb / 1e9, 3),
          "useful_frac": round(useful / PEAK_L2_GBPS, 4)}
    if lines:
        l2.update(line_gbyte_per_launch=round(lines / 1e9, 3), line_saturation=round(lines / (dom_ms * 1e-3) / 1e9 / PEAK_L2_GBPS, 4),
                  lines_per_useful_byte=round(lines / max(dom_tb, 1.0), 2))
    l2["note"] = ("what binds a table kernel: every 4-byte Gram gather that misses the L1 moves a 128-byte line (TCP_TCC_READ_REQ x 128 B "
                  "per launch / the launch time = line_saturation of the aggregate L2 -> L1 rate); useful_frac prices only the entries used")
    return {"bound": "hbm", "kernel": dom_name, "achieved": round(hbm_ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": round(hbm_ach / PEAK_HBM_GBPS, 4), "traffic": traffic, "traffic_source": pmc_src,
            "algorithmic_gbyte_per_launch": round(dom_by / 1e9, 3), "avg_launch_ms": round(float(dom_ms), 4), "l2": l2,
            "note": "a table kernel (no FLOPs): frac = algorithmic HBM bytes per launch (per-vector inputs, lists and tables written "
                    "and read back) / launch time (HIP events) / 8 TB/s -- the same definition every round; traffic = HBM-side bytes "
                    "of the PMC passes, (2*FETCH_SIZE + WRITE_SIZE) x 1024.  The kernel is not HBM-bound: see l2"}


def load_quantizer(state, D, K, N, dev):
    from quantization_amd import Quantizer
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd)
    return q.to(dev)


def timed(fn, reps):
    """seconds per call for the SECONDARY figures: median of `reps` individually synchronised calls after one untimed call
    (one host hiccup inside a three-call average once reported dim 256 / 4 codebooks at a fifth of its rate)"""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(max(reps, 3)):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(state, D, budget_s=30.0):
    """torch-CPU restatement of the reference's op sequence on this host: x ~ N(0,1) fp32 seed 0, chunk
    sweep {64, 256, 1024}, all host threads -- and, because torch on hundreds of threads is slower than on 8-32 for these small
    ops, the same chunks on 8 and 32 threads.  The whole sweep table is reported; `value` is a bounded sample (about budget_s of
    CPU work) at the best point, `cores` the threads it used."""
    from oracle.torch_port import TorchPortQuantizer
    ncpu = os.cpu_count() or 1
    port = TorchPortQuantizer(state)
    rs = np.random.RandomState(0)
    pool = torch.from_numpy(rs.standard_normal((1024, D)).astype(np.float32))
    torch.set_num_threads(min(ncpu, 8))
    port.encode(pool[:32], 5, chunk=32)   # warm-up (excluded)
    sweep, best = [], (0.0, min(ncpu, 8), 64)
    t_probe = time.time()
    for threads in sorted({min(ncpu, t) for t in (8, 32, ncpu)}):
        torch.set_num_threads(threads)
        for chunk in (64, 256, 1024):
            if time.time() - t_probe > 45:      # (a slow host: keep what the sweep has so far)
                break
            n = max(chunk, 128)
            t = time.time()
            port.encode(pool[:n], 5, chunk=chunk)
            r = n / (time.time() - t)
            sweep.append({"threads": threads, "chunk": chunk, "vectors": n, "vectors_per_s": round(r, 1)})
            if r > best[0]:
                best = (r, threads, chunk)
    rate, threads, chunk = best
    torch.set_num_threads(threads)
    n = int(max(chunk, min(32768, rate * budget_s)) // chunk * chunk)
    xs = torch.from_numpy(rs.standard_normal((n, D)).astype(np.float32))
    t = time.time()
    port.encode(xs, 5, chunk=chunk)
    dt = time.time() - t
    allc = [p_ for p_ in sweep if p_["threads"] == ncpu]
    return {"value": round(n / dt, 1), "unit": "vectors/s", "cores": threads, "kind": "port", "host_threads": ncpu,
            "sample": f"{n} Gaussian vectors (seed 0) of the same workload, torch-CPU restatement of the reference op "
                      f"sequence (oracle/torch_port.py), chunks of {chunk}, {threads} of {ncpu} host threads "
                      f"(the best point of the sweep), {dt:.1f} s",
            "all_threads_best": max((p_["vectors_per_s"] for p_ in allc), default=None),
            "sweep": sweep}


def fixture_parity(q, dev, iters):
    """HIP codes against the 4,096 rows the REFERENCE itself encoded (tests/golden/config_b_d512_n8.npz: data only)."""
    from quantization_amd import synthetic as gen
    path = os.path.join(ROOT, "tests", "golden", "config_b_d512_n8.npz")
    if not os.path.exists(path) or iters != 5:
        return None
    z = np.load(path)
    x = gen.make_gaussian(int(z["x_seed"]), int(z["B"]), int(z["D"]))
    assert gen.checksum(x) == float(z["x_checksum"])
    pinned = "scales_exp" in z.files
    if pinned:      # the two fp32 scale factors of the reference's run (torch's fp32 exp differs in the last bit between CPUs)
        q.pin_scale_factors(float(z["scales_exp"][0]), float(z["scales_exp"][1]))
    try:
        with torch.no_grad():
            got = q.encode(torch.from_numpy(x).to(dev), 5).cpu().numpy()
    finally:
        if pinned:
            q.pin_scale_factors()
    bad = (got != z["codes_it5"]).any(axis=1)
    margin = z["margin2_it5"] if "margin2_it5" in z.files else z["margin_it5"]
    return {"rows": int(len(bad)), "mismatches": int(bad.sum()),
            "near_tie_mismatches": int((bad & (margin < NEAR_TIE)).sum()),
            "clear_margin_mismatches": int((bad & (margin >= NEAR_TIE)).sum()),
            "note": "codes the reference's Quantizer.encode returned for the same seeded state and inputs; a near tie "
                    "has an fp64 decision margin < 2e-6 of the competing scores (the reference's own code flips under a re-ordered fp32 sum)"}


def oracle_pin(o_cls):
    """Which build of the CPU oracle checked this run (it is compiled where it runs): compiler, flags, a checksum of the library
    and a 64-row known-answer self-test against codes the REFERENCE produced (tests/golden/config_b_d512_n8.npz)."""
    import hashlib
    import re
    import subprocess
    from quantization_amd import synthetic as gen
    from oracle import oracle as omod
    info = {}
    try:
        mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        info["cflags"] = re.search(r"^CFLAGS\s*=\s*(.*)$", mk, re.M).group(1).strip()
        info["cc"] = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0]
        info["library_sha256_16"] = hashlib.sha256(open(omod._SO, "rb").read()).hexdigest()[:16]
        z = np.load(os.path.join(ROOT, "tests", "golden", "config_b_d512_n8.npz"))
        st = gen.synthetic_state(int(z["state_seed"]), int(z["D"]), int(z["K"]), int(z["N"]))
        o = o_cls(st["centers"], float(st["centers_scale"]), st["to_logits.weight"], st["to_logits.bias"], float(st["logits_scale"]),
                  scales_exp=(z["scales_exp"] if "scales_exp" in z.files else None))
        x = gen.make_gaussian(int(z["x_seed"]), int(z["B"]), int(z["D"]))[:64]
        got = o.encode(x, 5)
        info["self_test"] = {"rows": 64, "equal_to_reference_codes": bool(np.array_equal(got, z["codes_it5"][:64])),
                             "codes_sha256_16": hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()[:16]}
    except Exception as e:      # noqa: BLE001
        info["error"] = f"{type(e).__name__}: {e}"[:200]
    return info


def trainer_leg(dev, D, N, batch, p_iters, process_group=None, data_parallel=False, overlap=True, force_collectives=False):
    """ms per QuantizerTrainer.step in both phases (free-running, no host sync per step).  overlap=False: the whole gradient bucket
    in ONE all-reduce after the backward (MCQ_TRAINER_OVERLAP=0) instead of two overlapped parts."""
    from quantization_amd import QuantizerTrainer
    random.seed(0)
    torch.manual_seed(0)
    tr = QuantizerTrainer(dim=D, bytes_per_frame=N, device=dev, phase_one_iters=p_iters, phase_two_iters=p_iters,
                          process_group=process_group, data_parallel=data_parallel, force_collectives=force_collectives)
    tr.overlap_all_reduce = bool(overlap)
    xt = torch.randn(batch, D, device=dev)
    ms, nparam = {}, {}
    for phase, until in (("phase1", p_iters), ("phase2", 2 * p_iters + 1)):
        for _ in range(10):
            tr.step(xt)
        torch.cuda.synchronize()
        t4, n0 = time.perf_counter(), tr.cur_iter
        while tr.cur_iter < until - 5:
            tr.step(xt)
        torch.cuda.synchronize()
        ms[phase + "_ms_per_step"] = round((time.perf_counter() - t4) / (tr.cur_iter - n0) * 1e3, 3)
        qq = tr.quantizer
        # what a data-parallel step all-reduces: the flat gradient bucket + the forward batch sums (N K mean probabilities, N K counts, 4 sums)
        nparam[phase] = sum(p.numel() for p in qq.parameters()) + 2 * qq.num_codebooks * qq.codebook_size + 4
        while tr.cur_iter < until + (1 if until == p_iters else 0):
            tr.step(xt)
    return ms, nparam


_REAL_STDOUT = None


def emit(obj):
    """the JSON line, on the real stdout (see main)"""
    data = (json.dumps(obj) + "\n").encode()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        data = data[os.write(fd, data):]


def self_launch(n):
    """Re-run this command under torch.distributed.run with n ranks on 127.0.0.1 (a free port); returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, stdout=(_REAL_STDOUT if _REAL_STDOUT is not None else None))      # (the ranks' line goes to OUR real stdout)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--num-codebooks", type=int, default=8)
    ap.add_argument("--refine-iters", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass (for PMC runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a dry run)")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run of the N > 1 path on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (other configs, skipping mode, host-resident / fp16 input, decode, "
                         "trainer): every kernel launch of the run then has the headline shape (for rocprofv3 averages)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group, report the ranks seen and leave (covers the self-launch on a CPU-only box)")
    ap.add_argument("--no-pmc-check", action="store_true",
                    help="do not run the rocprofv3 --pmc child that reads the dominant kernel's L1 <- L2 requests live")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dp-iters", type=int, default=2000,
                    help="iterations per phase of the data-parallel config-E trainer leg under --gpus N > 1 (BASELINE config E: 10,000)")
    ap.add_argument("--shard-vectors", type=int, default=1 << 20,
                    help="vectors per GPU of the config-C shard leg under --gpus N > 1 (BASELINE config C: 8M over 8 GPUs)")
    args = ap.parse_args()

    # The ONE line of this command goes to the process's real stdout; everything else that writes to file descriptor 1 -- RCCL prints
    # a version banner there, from C, when its first communicator comes up, and it lands AFTER the line when the buffers are flushed at
    # exit -- is sent to stderr from here on.
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run
        # on the loopback address), pass every argument through and leave with the launcher's exit code
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_only:
        # the launch path alone (no device needed): every rank joins the group and reports in; rank 0 prints what it saw
        import torch.distributed as dist
        dist.init_process_group("gloo" if not torch.cuda.is_available() else args.backend)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "pid": os.getpid()})
        if rank == 0:
            emit({"rendezvous": world, "gpus": args.gpus, "backend": dist.get_backend(),
                  "ranks": sorted(r["rank"] for r in seen), "pids": len({r["pid"] for r in seen})})
        dist.barrier()
        dist.destroy_process_group()
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        return
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs an MI355X"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from quantization_amd import synthetic as gen
    from quantization_amd import _lib

    D, N, K, B, iters = args.dim, args.num_codebooks, 256, args.batch_per_gpu, args.refine_iters
    state = gen.synthetic_state(103, D, K, N)      # same seeded state as the config_b fixture
    q = load_quantizer(state, D, K, N, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(rank)
    x = torch.randn(B, D, generator=g, device=dev, dtype=torch.float32)   # resident in HBM

    def barrier():
        if dist is not None:
            dist.barrier()

    with torch.no_grad():
        for _ in range(args.warmup):
            codes = q.encode(x, iters)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            codes = q.encode(x, iters)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    per_rank_dt = [dt]
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_dt = [float(e.item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- BASELINE config C under --gpus N > 1: the 8M-vector encode sharded over the GPUs, 1,048,576 vectors per GPU, no collective on
    # the data path; every rank times its own shard between barriers, value = all shards / the slowest rank
    shard = None
    if dist is not None and not args.no_secondary:
        shard = {}
        try:
            Bs = args.shard_vectors
            gs = torch.Generator(device=dev)
            gs.manual_seed(1000 + rank)
            xs_ = torch.randn(Bs, D, generator=gs, device=dev, dtype=torch.float32)
            with torch.no_grad():
                q.encode(xs_, iters)
                torch.cuda.synchronize()
                barrier()
                torch.cuda.synchronize()
                t0s = time.perf_counter()
                for _ in range(2):
                    cs_ = q.encode(xs_, iters)
                torch.cuda.synchronize()
                barrier()
                torch.cuda.synchronize()
                ts = torch.tensor([time.perf_counter() - t0s], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(ts) for _ in range(world)]
            dist.all_gather(every, ts)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            shard = {"batch_per_gpu": Bs, "global_batch": Bs * world, "steps": 2, "encode_vectors_per_s": round(world * Bs * 2 / float(ts.item()), 1),
                     "ms_per_step": round(float(ts.item()) / 2 * 1e3, 2),
                     "per_rank_vectors_per_s": [round(Bs * 2 / float(e.item()), 1) for e in every],
                     "note": "BASELINE.json configs[2]: batch-sharded encode, 1,048,576 vectors per GPU (8,388,608 over 8 GPUs), no "
                             "collective on the data path; barrier + synchronize on both sides, the slowest rank's time"}
            del xs_, cs_
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001
            shard = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- data-parallel trainer (BASELINE config E): every rank takes part in the collectives.  Config E's schedule at
    # --dp-iters iterations per phase (default 2,000 + 2,000; the config's own 10,000 + 10,000 with --dp-iters 10000), for a global and
    # a per-GPU batch of 4,096 frames, with the gradient all-reduce in two overlapped parts and as one collective
    dp = None
    dp_failed = False
    if dist is not None and not args.no_secondary:
        dp = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "iterations_per_phase": args.dp_iters}
        try:      # (a failure here must not take the headline line with it)
            for tag, per_gpu in (("config_e_global_batch_4096", max(4096 // world, 64)), ("config_e_per_gpu_batch_4096", 4096)):
                for ov in (True, False):
                    ms, nparam = trainer_leg(dev, D, N, per_gpu, args.dp_iters, data_parallel=True, overlap=ov)
                    tt = torch.tensor([ms["phase1_ms_per_step"], ms["phase2_ms_per_step"]], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    d_ = dp.setdefault(tag, {"per_gpu_batch": per_gpu, "global_batch": per_gpu * world,
                                             "all_reduce_bytes_per_step_phase1": 4 * nparam["phase1"],
                                             "all_reduce_bytes_per_step_phase2": 4 * nparam["phase2"]})
                    key = "overlapped_two_part_all_reduce" if ov else "one_all_reduce"
                    d_[key] = {"phase1_ms_per_step": round(float(tt[0]), 3), "phase2_ms_per_step": round(float(tt[1]), 3),
                               "frames_per_s_phase2": round(per_gpu * world / (float(tt[1]) * 1e-3), 1)}
        except Exception as e:      # noqa: BLE001  (reported in the line AND in the exit code, after the line is out)
            dp["error"] = f"{type(e).__name__}: {e}"[:300]
            dp_failed = True
        dp["note"] = ("QuantizerTrainer.step, data_parallel=True: the flat gradient bucket all-reduced (RCCL) in two parts -- "
                      "the centers' gradient while the classifier's backward still runs, the rest after it -- or as one collective "
                      "(MCQ_TRAINER_OVERLAP=0), + one small forward all-reduce of the batch sums per step; max over ranks")

    if dist is not None:      # a failure of the trainer leg on ANY rank fails the run (after the headline line is printed)
        f = torch.tensor([1.0 if dp_failed else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        dp_failed = bool(f.item() > 0)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if dp_failed:
            sys.exit(3)
        return

    # ---- parity (outside the timed region): sampled rows vs the CPU oracle, fixture rows vs the reference
    from oracle.oracle import OracleQuantizer
    o = OracleQuantizer(state["centers"], float(state["centers_scale"]), state["to_logits.weight"],
                        state["to_logits.bias"], float(state["logits_scale"]))
    rows = np.random.RandomState(1).choice(B, min(256, B), replace=False)
    want = o.encode(x[rows].cpu().numpy(), iters)
    parity = {"sampled_rows_vs_oracle": int(len(rows)), "bit_exact": bool(np.array_equal(codes[rows].cpu().numpy(), want)),
              "oracle_build": oracle_pin(OracleQuantizer)}
    if (D, N, K) == (512, 8, 256) and not args.no_secondary:      # (--no-secondary: every launch has the headline shape)
        parity["vs_reference_fixture"] = fixture_parity(q, dev, iters)

    # ---- per-kernel HIP-event timing on the launch stream (same inputs, same process): mcq_profile_encode enqueues exactly
    # what Quantizer.encode enqueues, with an event pair round the launches of one category per profiled encode
    L = _lib.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    kernels = {} if args.no_profile else profile_kernels(L, q, x, B, D, N, K, iters, dev)
    kernels_sum_ms = round(float(sum(v["ms_per_encode"] for v in kernels.values())), 3)
    # the dominant kernel = the category with the largest time per encode (what rocprofv3 --stats puts on top).  Its counters:
    # read LIVE by a rocprofv3 --pmc child of this very command (one pass per counter group, as MI355X_MICROARCH.md prescribes:
    # counters cannot be read from inside the timed process), else from the newest committed passes of the same command
    roofline = None
    pmc_all = None
    if kernels:
        dom_name = max(kernels, key=lambda n: kernels[n]["ms_per_encode"])
        pmc, pmc_src = None, None
        if world == 1 and not args.no_pmc_check and not args.no_secondary:
            try:      # (a rocprofv3 format change or a partial output file must not take the headline line with it)
                pmc_all = pmc_live(D, N, B, iters)
            except Exception as e:      # noqa: BLE001
                pmc_all = None
                sys.stderr.write(f"bench.py: live --pmc read failed ({type(e).__name__}: {e}); falling back to the committed passes\n")
            if pmc_all and dom_name in pmc_all:
                pmc, pmc_src = pmc_all[dom_name], "live: rocprofv3 --kernel-trace --pmc child runs of this command (bench.py pmc_live)"
        if pmc is None:
            pmc, pmc_src = pmc_committed(dom_name, D, N, K, B, iters)
        roofline = roofline_of(dom_name, kernels[dom_name], kernel_work(kernels[dom_name]["vectors_per_launch"], D, N, K)[dom_name], pmc, pmc_src)

    fpv = reference_flops_per_vector(D, N, K, iters)
    exec_fpv = 2 * 2.0 * D * N * K          # the logits and x.C products only (each multiply-add = ten i8 limb products)
    _w = kernel_work(B, D, N, K)
    _pass_names = ("stage0_tables", "combine_level0", "level1_combines_and_tables", "combine_level2", "tables_upper_levels") if N >= 8 \
        else ("stage0_tables", "combine_level0", "combine_level1")
    executed_floor_ms = round((LIMB_PRODUCTS * exec_fpv * B / (PEAK_I8_MFMA_TOPS * 1e12) +
                               iters * sum(_w[n_][2] + (_w[n_][1] if n_ == "stage0_tables" else 0.0) for n_ in _pass_names) / (PEAK_L2_GBPS * 1e9)) * 1e3, 3)
    # the same floor with the table passes priced by the 128-byte LINES their gathers move from L2 to L1 (what the L2 serves:
    # one line per channel and clock whatever part of it is used), from the committed --pmc passes of this workload
    line_floor_ms, line_floor_src = None, None
    _pmc = pmc_all
    if _pmc is not None:
        line_floor_src = "live"
    elif (D, N, K, B, iters) == (512, 8, 256, 65536, 5):
        import glob as _glob2
        _pf = sorted(_glob2.glob(os.path.join(ROOT, "profiles", "r??_pmc_traffic.json")))
        if _pf:
            _pmc, line_floor_src = json.load(open(_pf[-1])), "committed: profiles/" + os.path.basename(_pf[-1])
    if _pmc:
        _lines = sum((_pmc.get(n_, {}).get("l2_read_request_bytes") or 0) for n_ in _pass_names)
        if _lines > 0:
            line_floor_ms = round((LIMB_PRODUCTS * exec_fpv * B / (PEAK_I8_MFMA_TOPS * 1e12) + iters * _lines / (PEAK_L2_GBPS * 1e9)) * 1e3, 3)
    value = world * B * args.steps / dt
    if roofline is not None:
        roofline["whole_encode"] = {"ms_per_step": round(dt / args.steps * 1e3, 3), "executed_floor_ms": executed_floor_ms,
                                    "frac_of_floor": round(executed_floor_ms / (dt / args.steps * 1e3), 4),
                                    "line_floor_ms": line_floor_ms,
                                    "frac_of_line_floor": None if line_floor_ms is None else round(line_floor_ms / (dt / args.steps * 1e3), 4),
                                    "note": "the whole encode against what THIS algorithm costs at the chip's peaks (products at the dense i8 "
                                            "peak + the table passes' useful bytes at the L2 peak), and against the same with the table passes "
                                            "priced by the 128-byte lines they move: see whole_encode"}
    out = {
        "metric": "vectors encoded/sec at dim=512, 8 codebooks; uint8 codes bit-exact vs ref",
        "value": round(value, 1), "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (products: exact 30-bit fixed point on the i8 MFMA)", "data": "synthetic",
        "config": {"workload": f"Quantizer.encode, dim={D}, bytes_per_frame={N}, codebook_size={K}, "
                               f"refine_indexes_iters={iters}, batch={B} fp32 Gaussian vectors per GPU "
                               f"(BASELINE.json configs[1]), seeded synthetic codebooks",
                   "global_batch": world * B, "parallelism": f"batch-sharded x{world}, no collective"},
        "ranks": {"rccl_ranks": world, "backend": (dist.get_backend() if dist is not None else "none (one process)"),
                  "launcher": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "plain python",
                  "per_rank_vectors_per_s": [round(B * args.steps / t_, 1) for t_ in per_rank_dt],
                  "note": "value = all ranks' vectors / the slowest rank's time (barrier + synchronize on both sides)"},
        "parity": parity,
        "whole_encode": {"kernels_sum_ms": kernels_sum_ms,
                         "executed_floor_ms": executed_floor_ms, "frac_of_floor": round(executed_floor_ms / (dt / args.steps * 1e3), 4),
                         "line_floor_ms": line_floor_ms, "line_floor_counters": line_floor_src,
                         "frac_of_line_floor": None if line_floor_ms is None else round(line_floor_ms / (dt / args.steps * 1e3), 4),
                         "line_floor_note": "as executed_floor_ms, but the table passes move the 128-byte lines their 4-byte gathers "
                                            "touch (TCP_TCC_READ_REQ x 128 B per launch, profiles/rNN_pmc_traffic.json) at the L2 peak: "
                                            "the floor of THIS data layout (fp32 Gram matrix, 16 of a segment's 256 entries per row)",
                         "executed_floor_note": "products: 2 x 2*D*N*K multiply-adds x 10 limb products per vector at the dense i8 peak; "
                                                "table passes: the useful Gram / x.C bytes of the five table kernels at the L2 peak "
                                                "(34.5 TB/s); what an encode of this ALGORITHM costs at the chip's peaks",
                         "reference_flop_per_vector": fpv, "executed_product_flop_per_vector": exec_fpv,
                         "executed_i8_mfma_op_per_vector": LIMB_PRODUCTS * exec_fpv,
                         "reference_tflops": round(value / world * fpv / 1e12, 2),
                         "frac_of_f32_mfma_peak": round(value / world * fpv / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "note": "reference_* prices the matmul FLOPs of the reference algorithm (SURVEY.md 8d: 6.12 M "
                                 "vectors/s at the fp32-MFMA peak); the table form executes only the logits and x.C products -- as exact "
                                 "fixed-point products on the i8 matrix cores -- and reads the other inner products from the "
                                 "Gram matrix, so this fraction can exceed 1"},
        "roofline": roofline,
        "kernels": kernels,
    }
    if dp is not None:
        out["dp_trainer"] = dp
    if shard is not None:
        out["configs"] = {"C_shard_dim512_bytes8_1M": shard}
    if args.no_secondary:
        emit(out)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- decode (gather-sum, HBM-write-bound by its algorithmic bytes N + 4*D per vector): back-to-back mcq_decode
    # launches through the C ABI into one output buffer, HIP events on the launch stream around the burst
    def decode_burst(qq, cc, n_, d_, reps=50):
        """ms per mcq_decode launch: `reps` back-to-back launches through the C ABI into one output buffer"""
        with torch.no_grad():
            y = qq.decode(cc)
            dblob = qq._prepared(any_flavour=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(4 * reps):  # untimed: a burst this short otherwise runs before the clocks have come up
                L.mcq_decode(cc.data_ptr(), 1, n_, cc.shape[0], dblob.data_ptr(), n_, 256, d_, y.data_ptr(), st)
            e0.record()
            for _ in range(reps):
                rc = L.mcq_decode(cc.data_ptr(), 1, n_, cc.shape[0], dblob.data_ptr(), n_, 256, d_, y.data_ptr(), st)
                assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            assert torch.equal(y, qq.decode(cc))
            return e0.elapsed_time(e1) / reps

    dec_ms = decode_burst(q, codes, N, D)
    dec_gbps = B * (N + 4 * D) / (dec_ms * 1e-3) / 1e9
    out["decode"] = {"vectors_per_s": round(B / (dec_ms * 1e-3), 1), "ms": round(dec_ms, 4),
                     "hbm_gb_per_s": round(dec_gbps, 1), "peak_gb_per_s": PEAK_HBM_GBPS, "frac": round(dec_gbps / PEAK_HBM_GBPS, 4),
                     "note": "50 back-to-back mcq_decode launches (after 200 untimed ones), HIP events on the launch stream; algorithmic bytes = N + 4*D per vector"}

    # ---- the other BASELINE shapes on one GPU (same code path; parity for them is in tests/ -m gpu)
    if world == 1:
        cfgs = {}
        for name, d_, n_, b_, reps in (("A_dim256_bytes4", 256, 4, 65536, 3), ("D_dim1024_bytes16", 1024, 16, 65536, 2),
                                        ("C_shard_dim512_bytes8_1M", 512, 8, 1 << 20, 1)):
            qc = q if (d_, n_) == (D, N) else load_quantizer(gen.synthetic_state(103, d_, 256, n_), d_, 256, n_, dev)
            xc_ = torch.randn(b_, d_, device=dev)
            with torch.no_grad():
                t_enc = timed(lambda: qc.encode(xc_, 5), reps)
                cc = qc.encode(xc_, 5)
                t_dec = decode_burst(qc, cc, n_, d_, 20 if b_ <= 65536 else 5) * 1e-3
            f_ = reference_flops_per_vector(d_, n_, 256, 5)
            cfgs[name] = {"batch": b_, "encode_vectors_per_s": round(b_ / t_enc, 1), "encode_ms": round(t_enc * 1e3, 2),
                          "frac_of_f32_mfma_peak_reference_flops": round(b_ / t_enc * f_ / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                          "decode_gb_per_s": round(b_ * (n_ + 4 * d_) / t_dec / 1e9, 1),
                          "decode_frac": round(b_ * (n_ + 4 * d_) / t_dec / 1e9 / PEAK_HBM_GBPS, 4)}
            if b_ <= 65536 and not args.no_profile:
                # the shape's own dominant launch and what bounds it (SURVEY 8d for the named shapes, not only the headline one):
                # launch times measured here; counters from the committed --pmc passes of that shape (profiles/r05_config_*)
                kk = profile_kernels(L, qc, xc_, b_, d_, n_, 256, 5, dev, reps=1)
                dn = max(kk, key=lambda n__: kk[n__]["ms_per_encode"])
                pm, pm_src = pmc_committed(dn, d_, n_, 256, b_, 5)
                cfgs[name]["roofline"] = roofline_of(dn, kk[dn], kernel_work(kk[dn]["vectors_per_launch"], d_, n_, 256)[dn], pm, pm_src)
                cfgs[name]["kernels_ms_per_encode"] = {k_: v_["ms_per_encode"] for k_, v_ in kk.items()}
            del xc_, cc
            if qc is not q:
                del qc
            torch.cuda.empty_cache()
        out["configs"] = cfgs

    # ---- secondary: the same encode with fixed-point skipping (identical codes, data-dependent cost;
    # never the headline value: BASELINE's metric is the reference's fixed 5-pass work)
    with torch.no_grad():
        q.skip_fixed_points = True
        c2 = q.encode(x, iters)
        skip_dt = timed(lambda: q.encode(x, iters), 3)
        q.skip_fixed_points = False
    out["fixed_point_skipping"] = {"vectors_per_s": round(B / skip_dt, 1), "ms_per_step": round(skip_dt * 1e3, 3),
                                   "codes_identical": bool(torch.equal(c2, codes)),
                                   "note": "opt-in Quantizer.skip_fixed_points: converged vectors leave later passes"}

    # ---- secondary: batch resident in (pinned) HOST memory, H2D copies overlapped with the kernels
    with torch.no_grad():
        xh = torch.cat([x.cpu(), x.cpu()]).pin_memory()
        ch = q.encode_from_host(xh, iters, chunk=B // 2)          # untimed: staging buffers of this chunk size, copy stream
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(3):
            ch = q.encode_from_host(xh, iters, chunk=B // 2)      # 4 chunks of B/2
        host_dt = (time.perf_counter() - t2) / 3
    out["host_resident_input"] = {"vectors_per_s": round(2 * B / host_dt, 1),
                                  "codes_identical": bool(torch.equal(ch[:B], codes.cpu())),
                                  "note": "PCIe-inclusive (pinned host batch of 2x65,536 vectors, double-buffered "
                                          "H2D on a copy stream); never the headline value"}
    del xh

    # ---- secondary: fp16 frames resident in HBM, widened in the kernels' load path (same codes by construction)
    with torch.no_grad():
        x16 = x.to(torch.float16)
        c16 = q.encode(x16, iters)
        t16 = timed(lambda: q.encode(x16, iters), 3)
        out["fp16_input"] = {"vectors_per_s": round(B / t16, 1),
                             "codes_equal_widened_input": bool(torch.equal(c16, q.encode(x16.float(), iters)))}

    # ---- secondary: latency of small encodes (a serving-sized request: the launch chain, not the kernels, is what it waits for)
    with torch.no_grad():
        lat = {}
        for bs in (64, 4096):
            xs = x[:bs].contiguous()
            q.encode(xs, iters)
            lat["batch_%d_ms" % bs] = round(timed(lambda: q.encode(xs, iters), 50) * 1e3, 4)
        lat["note"] = "Quantizer.encode of one small batch, 5 passes, median of 50 synchronised calls (host launch + ~30 dependent kernels)"
        out["small_batch_latency"] = lat

    # ---- secondary: QuantizerTrainer.step (BASELINE config E shape on one GPU: dim 512, 8 bytes, batch 4096)
    if world == 1:
        try:
            ms_t, _ = trainer_leg(dev, D, N, 4096, 60)
            out["trainer_step"] = dict(ms_t, batch=4096, note="QuantizerTrainer.step, fused (autograd-free) path, free-running")
            # the same step with every collective of the data-parallel path issued through RCCL in a group of ONE rank
            # (force_collectives): what the collectives cost on this box before any link is involved
            try:
                import socket
                import torch.distributed as dist1
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    port = so.getsockname()[1]
                dist1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
                one = {}
                for ov in (True, False):
                    ms1, np1 = trainer_leg(dev, D, N, 4096, 60, data_parallel=True, overlap=ov, force_collectives=True)
                    one["overlapped_two_part_all_reduce" if ov else "one_all_reduce"] = ms1
                one["all_reduce_bytes_per_step"] = {k_: 4 * v_ for k_, v_ in np1.items()}
                one["note"] = ("one-rank nccl (= RCCL) group, force_collectives=True: the broadcast of the parameters, the forward all-reduce "
                               "and the gradient all-reduce(s) of every step run through RCCL; against trainer_step.phase*_ms_per_step "
                               "this is the collectives' fixed cost")
                out["trainer_step"]["one_rank_rccl"] = one
                dist1.destroy_process_group()
            except Exception as e:      # noqa: BLE001
                out["trainer_step"]["one_rank_rccl"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            # BASELINE config E at its length: the trainer's DEFAULT schedule (10,000 + 10,000 iterations, quantization.py:581-583),
            # batches of 4,096 fresh Gaussian frames (tests/test_gpu_trainer_long.py checks what such a run converges to)
            from quantization_amd import QuantizerTrainer
            import gc
            gc.collect()
            gc.freeze()         # (the step is host-bound: a generation-2 collection over everything this process has built costs it 5 %)
            torch.cuda.empty_cache()
            random.seed(0)
            torch.manual_seed(0)
            tr = QuantizerTrainer(dim=D, bytes_per_frame=N, device=dev)
            gq = torch.Generator(device=dev)
            gq.manual_seed(1)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            nsteps = 0
            while not tr.done():
                tr.step(torch.randn(4096, D, device=dev, generator=gq))
                nsteps += 1
            torch.cuda.synchronize()
            tot = time.perf_counter() - t5
            gc.unfreeze()
            out["trainer_step"]["config_e"] = {"steps": nsteps, "trainer_total_s": round(tot, 2), "ms_per_step": round(tot / nsteps * 1e3, 4),
                                               "frames_per_s": round(nsteps * 4096 / tot, 1),
                                               "note": "QuantizerTrainer(dim=512, bytes_per_frame=8) with its default 10,000 + 10,000 iterations "
                                                       "on one GPU, 4,096 frames per step, frame generation included; gc.freeze() before the leg "
                                                       "(the step is host-bound, and Python's generation-2 collections over everything the earlier legs "
                                                       "built cost it 5 %: a stand-alone script measures the same as this, tools/exp_config_e_after_rccl.py)"}
        except Exception as e:      # noqa: BLE001  (an optional leg: the headline line must still come out)
            out.setdefault("trainer_step", {})["error"] = f"{type(e).__name__}: {e}"[:300]
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(state, D)
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "port", "sample": "failed", "error": f"{type(e).__name__}: {e}"[:300]}
    emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if dp_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
