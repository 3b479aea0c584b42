#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh.
usage: pmc_traffic.py gpurun_out/<pmc dir> > profiles/rNN_pmc_traffic.json   (keys = bench.py's kernel categories)"""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
KERNELS = {"logits_product_argmax": "k_fgemm<1>", "xc_product": "k_fgemm<0>", "frames_to_limbs": "k_fix_rows<4>",
           "residual_energies": "k_tf_er<8>", "stage0_tables": "k_tf_stage0<256, 8>", "combine_level0": "k_tf_pair0<16>",
           "combine_level1": "k_tf_pair1<16, 16>", "tables_level1": "k_tf_table1<16, 16>", "combine_level2": "k_tf_comb<16, 32, true>",
           # (outside mcq_profile_encode the level-1 combines and the level-2 cousin tables share one launch)
           "level1_combines_and_tables": "k_tf_level1<16, 16>"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum"):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mcq::", "")
            n = n.replace(", unsigned char>", ">")      # (the entry type CT of the table kernels: one byte on this workload)
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per launch, mean over launches) for bench.py's "
                "default workload (dim 512, 8 codebooks, 65,536 vectors; bench.py --no-secondary: every launch has that shape); "
                "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- gfx950 FETCH_SIZE reads 1/2 of a wide streaming read "
                "(MI355X_MICROARCH.md, HBM).  Counts fabric-side requests of the XCDs' L2s, so Infinity-Cache hits "
                "are included; Gram-table reads served by an L2 are not.",
       "source": "profiles/rNN_pmc_counters.txt (tools/pmc_passes.sh, tools/pmc_traffic.py)"}
for cat, k in KERNELS.items():
    f, w = vals[k]["FETCH_SIZE"], vals[k]["WRITE_SIZE"]
    if not (f and w):
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[cat] = {"kernel": k, "launches_averaged": len(f), "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                "traffic_bytes": int(round((2 * fk + wk) * 1024))}
    rq = vals[k]["TCP_TCC_READ_REQ_sum"]
    if rq:      # L1 -> L2 read requests, 128 bytes each (stage 0's coalesced 4.7 GB arrive in 33.4 M requests: 141 B per request)
        out[cat]["l2_read_requests"] = int(round(sum(rq) / len(rq)))
        out[cat]["l2_read_request_bytes"] = int(round(sum(rq) / len(rq) * 128))
print(json.dumps(out, indent=1))
