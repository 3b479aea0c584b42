#!/usr/bin/env python
"""profiles/r01_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh.
usage: pmc_traffic.py gpurun_out/<pmc dir> > profiles/r01_pmc_traffic.json"""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
KERNELS = {"logits_argmax": "k_gemm8s<16, 0, 4>", "residual": "k_residual_reg<8, 2>", "stage0_gemm": "k_gemm8s<16, 3, 4>",
           "pair_L1_K16": "k_pair<1, 16, true, 0, false>", "pair_L2_K16": "k_pair<2, 16, false, 0, false>", "pair_L4_K32": "k_pair<4, 32, false, 0, true>"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mcq::", "")
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per launch, mean over launches) for bench.py's "
                "default workload (dim 512, 8 codebooks, 65,536 vectors; bench.py --no-secondary: every launch has that shape); "
                "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- gfx950 FETCH_SIZE reads 1/2 of a wide streaming read "
                "(MI355X_MICROARCH.md, HBM); WRITE_SIZE was checked against the known 537 MB S0 store of the unfused "
                "stage0_gemm (524,288 KB).  Counts fabric-side requests, so Infinity-Cache hits (the gathered codebook rows) "
                "are included.",
       "source": "profiles/r01_pmc_counters.txt (tools/pmc_passes.sh, tools/pmc_traffic.py)"}
for cat, k in KERNELS.items():
    f, w = vals[k]["FETCH_SIZE"], vals[k]["WRITE_SIZE"]
    assert f and w, (cat, k, sorted(vals))
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[cat] = {"kernel": k, "launches_averaged": len(f), "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                "traffic_bytes": int(round((2 * fk + wk) * 1024))}
print(json.dumps(out, indent=1))
