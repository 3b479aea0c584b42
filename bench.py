#!/usr/bin/env python
"""Throughput of the encode hot path on MI355X: vectors encoded per second at
dim=512, 8 codebooks of 256 entries, 5 refinement passes (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one Quantizer.encode() of the per-GPU batch (65,536 synthetic Gaussian
vectors already resident in HBM; uint8 codes written to HBM).  With N GPUs every rank
encodes its own shard -- no collective on the data path (weak scaling).  Rank 0 prints
ONE JSON line.  `roofline` is for the dominant kernel, from per-launch HIP events on the
launch stream (mcq_profile_encode); `cpu_baseline` times the torch-CPU restatement of the
reference's op sequence (oracle/torch_port.py) on this host's cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def k_cutoff(K, L):
    kc = 8 if K <= 16 else 16
    while L >= 4:
        L //= 4
        kc *= 2
    return min(kc, 128)


def kernel_flops(B, D, N, K):
    """matmul FLOPs (2*m*n*k) of ONE launch of each kernel category, SURVEY.md 8(d)."""
    cats = [("logits_argmax", 2.0 * D * N * K * B), ("residual", 0.0), ("stage0_gemm", 2.0 * D * N * K * B),
            ("prune0", 0.0)]
    G, L, KI = N, 1, (1 if N == 1 else k_cutoff(K, 1))
    while G > 1:
        Gout = G // 2
        cats.append((f"pair_L{L}_K{KI}", Gout * KI * KI * 2.0 * D * B))
        KI = 1 if Gout == 1 else k_cutoff(K, 2 * L)
        G, L = Gout, 2 * L
    return cats


def total_flops_per_vector(D, N, K, iters):
    cats = kernel_flops(1, D, N, K)
    return cats[0][1] + iters * sum(f for _, f in cats[1:])


def cpu_baseline(state, D, budget_s=12.0):
    """torch-CPU restatement of the reference's op sequence on this host: a short probe picks
    the thread count and chunk size (torch on hundreds of threads is slower than on 16-32 for
    these small ops), then a bounded sample is timed."""
    from oracle.torch_port import TorchPortQuantizer
    ncpu = os.cpu_count() or 1
    port = TorchPortQuantizer(state)
    rs = np.random.RandomState(99)
    probe = torch.from_numpy(rs.standard_normal((128, D)).astype(np.float32))
    torch.set_num_threads(min(ncpu, 8))
    port.encode(probe[:32], 5, chunk=32)   # warm-up
    best = (0.0, min(ncpu, 8), 128)
    t_probe = time.time()
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(threads)
        for chunk in (64, 128):
            t = time.time()
            port.encode(probe, 5, chunk=chunk)
            r = 128 / (time.time() - t)
            if r > best[0]:
                best = (r, threads, chunk)
        if time.time() - t_probe > 20:
            break
    rate, threads, chunk = best
    torch.set_num_threads(threads)
    n = int(max(chunk, min(16384, rate * budget_s)) // chunk * chunk)
    xs = torch.from_numpy(rs.standard_normal((n, D)).astype(np.float32))
    t = time.time()
    port.encode(xs, 5, chunk=chunk)
    dt = time.time() - t
    return {"value": round(n / dt, 1), "unit": "vectors/s", "cores": threads, "kind": "port",
            "sample": f"{n} Gaussian vectors of the same workload, torch-CPU restatement of the reference op "
                      f"sequence (oracle/torch_port.py), chunks of {chunk}, {threads} of {ncpu} host threads "
                      f"(best of a short sweep), {dt:.1f} s"}


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
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (skipping mode, host-resident / fp16 input, decode): every kernel "
                         "launch of the run then has the headline shape (for rocprofv3 averages)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from golden import gen
    from quantization_amd import Quantizer, _lib

    D, N, K, B, iters = args.dim, args.num_codebooks, 256, args.batch_per_gpu, args.refine_iters
    state = gen.synthetic_state(103, D, K, N)      # same seeded state as the config_b fixture
    q = Quantizer(D, K, N)
    sd = q.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd)
    q = q.to(dev)
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
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- parity spot check (outside the timed region): sampled rows vs the CPU oracle
    from oracle.oracle import OracleQuantizer
    o = OracleQuantizer(state["centers"], float(state["centers_scale"]), state["to_logits.weight"],
                        state["to_logits.bias"], float(state["logits_scale"]))
    rows = np.random.RandomState(1).choice(B, 256, replace=False)
    want = o.encode(x[rows].cpu().numpy(), iters)
    parity_ok = bool(np.array_equal(codes[rows].cpu().numpy(), want))

    # ---- per-kernel HIP-event timing on the launch stream (same inputs, same process)
    L = _lib.lib()
    with torch.no_grad():          # inference flavour of the derived state (host-side scale factors)
        blob = q._prepared()
    ws = q._workspace(B, dev)
    ms = (ctypes.c_float * 32)()
    acc = np.zeros(32)
    reps = 3
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(0 if args.no_profile else reps):
        ncat = L.mcq_profile_encode(x.data_ptr(), B, blob.data_ptr(), q._lscale_exp, N, K, D, iters, ws.data_ptr(),
                                    ws.numel(), st, ms, 32)
        assert ncat > 0, ncat
        acc[:ncat] += np.array(ms[:ncat])
    acc /= reps
    acc = np.maximum(acc, 1e-9)
    cats = kernel_flops(B, D, N, K)
    kernels = {}
    for i, (name, fl) in enumerate(cats):
        launches = 1 if i == 0 else iters
        avg_ms = acc[i] / launches
        kernels[name] = {"launches_per_encode": launches, "avg_ms": round(float(avg_ms), 4),
                         "gflop_per_launch": round(fl / 1e9, 2),
                         "tflops": round(fl / (avg_ms * 1e-3) / 1e12, 2) if avg_ms > 0 and fl > 0 else 0.0,
                         "ms_per_encode": round(float(acc[i]), 3)}
    dom = max(range(len(cats)), key=lambda i: acc[i])
    dom_name, dom_fl = cats[dom]
    dom_ms = acc[dom] / (1 if dom == 0 else iters)
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_fl > 0 else 0.0

    # HBM-side traffic of the dominant kernel: from the committed rocprofv3 --pmc passes of this same
    # workload (counters cannot be collected from inside the timed process); null for other shapes
    traffic, traffic_note = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if (D, N, K, B, iters) == (512, 8, 256, 65536, 5) and os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if dom_name in pmc:
            traffic = pmc[dom_name]["traffic_bytes"]
            traffic_note = "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/r01_pmc_traffic.json"

    fpv = total_flops_per_vector(D, N, K, iters)
    value = world * B * args.steps / dt
    out = {
        "metric": "vectors encoded/sec at dim=512, 8 codebooks; uint8 codes bit-exact vs ref",
        "value": round(value, 1), "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Quantizer.encode, dim={D}, bytes_per_frame={N}, codebook_size={K}, "
                               f"refine_indexes_iters={iters}, batch={B} fp32 Gaussian vectors per GPU "
                               f"(BASELINE.json configs[1]), seeded synthetic codebooks",
                   "global_batch": world * B, "parallelism": f"batch-sharded x{world}, no collective"},
        "parity": {"sampled_rows_vs_oracle": 256, "bit_exact": parity_ok},
        "whole_encode": {"flop_per_vector": fpv, "tflops": round(value / world * fpv / 1e12, 2),
                         "frac_of_f32_mfma_peak": round(value / world * fpv / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
        "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2),
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                     "measured_issue_ceiling": 152.5,   # TFLOP/s: 33 cycles per v_mfma_f32_16x16x4_f32 at 2.4 GHz (tools/micro/mfma_chain.hip)
                     "gflop_per_launch": round(dom_fl / 1e9, 2),
                     "avg_launch_ms": round(float(dom_ms), 4)},
        "kernels": kernels,
    }
    if args.no_secondary:
        print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- secondary: the same encode with fixed-point skipping (identical codes, data-dependent cost;
    # never the headline value: BASELINE's metric is the reference's fixed 5-pass work)
    with torch.no_grad():
        q.skip_fixed_points = True
        c2 = q.encode(x, iters)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            c2 = q.encode(x, iters)
        torch.cuda.synchronize()
        skip_dt = (time.perf_counter() - t1) / 3
        q.skip_fixed_points = False
    out["fixed_point_skipping"] = {"vectors_per_s": round(B / skip_dt, 1), "ms_per_step": round(skip_dt * 1e3, 3),
                                   "codes_identical": bool(torch.equal(c2, codes)),
                                   "note": "opt-in Quantizer.skip_fixed_points: converged vectors leave later passes"}

    # ---- secondary: batch resident in (pinned) HOST memory, H2D copies overlapped with the kernels
    with torch.no_grad():
        xh = torch.cat([x.cpu(), x.cpu()]).pin_memory()
        q.encode_from_host(xh[:8192], iters, chunk=4096)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(2):
            ch = q.encode_from_host(xh, iters, chunk=B // 2)      # 4 chunks of B/2
        host_dt = (time.perf_counter() - t2) / 2
    out["host_resident_input"] = {"vectors_per_s": round(2 * B / host_dt, 1),
                                  "codes_identical": bool(torch.equal(ch[:B], codes.cpu())),
                                  "note": "PCIe-inclusive (pinned host batch of 2x65,536 vectors, double-buffered "
                                          "H2D on a copy stream); never the headline value"}

    # ---- secondary: fp16 frames resident in HBM, widened in the kernels' load path (same codes by construction)
    with torch.no_grad():
        x16 = x.to(torch.float16)
        c16 = q.encode(x16, iters)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(3):
            q.encode(x16, iters)
        torch.cuda.synchronize()
        out["fp16_input"] = {"vectors_per_s": round(3 * B / (time.perf_counter() - t3), 1),
                             "codes_equal_widened_input": bool(torch.equal(c16, q.encode(x16.float(), iters)))}

    # ---- decode (HBM-write-bound gather-sum), secondary figure
    with torch.no_grad():
        for _ in range(2):
            y = q.decode(codes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = q.decode(codes)
        e1.record()
        torch.cuda.synchronize()
        dec_ms = e0.elapsed_time(e1) / 10
    out["decode"] = {"vectors_per_s": round(B / (dec_ms * 1e-3), 1), "ms": round(dec_ms, 4),
                     "hbm_gb_per_s": round(B * (N + 4 * D) / (dec_ms * 1e-3) / 1e9, 1), "peak_gb_per_s": 8000.0,
                     "note": "torch current stream; algorithmic bytes = N + 4*D per vector"}
    # ---- secondary: QuantizerTrainer.step (BASELINE config E shape on one GPU: dim 512, 8 bytes, batch 4096)
    if world == 1:
        import random
        from quantization_amd import QuantizerTrainer
        random.seed(0)
        torch.manual_seed(0)
        tr = QuantizerTrainer(dim=D, bytes_per_frame=N, device=dev, phase_one_iters=60, phase_two_iters=60)
        xt = torch.randn(4096, D, device=dev)
        ms = {}
        for phase, until in (("phase1_ms_per_step", 60), ("phase2_ms_per_step", 121)):
            for _ in range(10):
                tr.step(xt)
            torch.cuda.synchronize()
            t4, n0 = time.perf_counter(), tr.cur_iter
            while tr.cur_iter < until - 5:
                tr.step(xt)
            torch.cuda.synchronize()
            ms[phase] = round((time.perf_counter() - t4) / (tr.cur_iter - n0) * 1e3, 3)
            while tr.cur_iter < until + (1 if until == 60 else 0):
                tr.step(xt)
        out["trainer_step"] = dict(ms, batch=4096, note="QuantizerTrainer.step, fused (autograd-free) path, free-running")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(state, D)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
