#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh.
usage: pmc_traffic.py gpurun_out/<pmc dir> [dim codebooks [batch passes]] > profiles/rNN_pmc_traffic.json   (keys = bench.py's kernel categories)"""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
DIM, NCB = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 8)
BATCH, ITERS = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (65536, 5)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import category_kernels      # category -> kernel-name prefix (the same map bench.py's live read uses)
KERNELS = category_kernels(NCB)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum"):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mcq::", "")
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per launch, mean over launches) for bench.py's "
                "default workload (dim 512, 8 codebooks, 65,536 vectors; bench.py --no-secondary: every launch has that shape); "
                "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- gfx950 FETCH_SIZE reads 1/2 of a wide streaming read "
                "(MI355X_MICROARCH.md, HBM).  Counts fabric-side requests of the XCDs' L2s, so Infinity-Cache hits "
                "are included; Gram-table reads served by an L2 are not.",
       "source": "profiles/rNN_pmc_counters.txt (tools/pmc_passes.sh, tools/pmc_traffic.py)"}
out["_note"] = out["_note"].replace("dim 512, 8 codebooks", f"dim {DIM}, {NCB} codebooks")
# the workload these counters belong to: bench.py's pmc_committed() only prices a launch of EXACTLY this shape with them
out["_shape"] = {"D": DIM, "N": NCB, "K": 256, "B": BATCH, "iters": ITERS}
for cat, prefix in KERNELS.items():
    names = [n for n in vals if n.startswith(prefix)]
    f = [x for n in names for x in vals[n]["FETCH_SIZE"]]
    w = [x for n in names for x in vals[n]["WRITE_SIZE"]]
    if not (f and w):
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[cat] = {"kernel": names[0], "launches_averaged": len(f), "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                "traffic_bytes": int(round((2 * fk + wk) * 1024))}
    rq = [x for n in names for x in vals[n]["TCP_TCC_READ_REQ_sum"]]
    if rq:      # L1 -> L2 read requests, 128 bytes each (stage 0's coalesced 4.7 GB arrive in 33.4 M requests: 141 B per request)
        out[cat]["l2_read_requests"] = int(round(sum(rq) / len(rq)))
        out[cat]["l2_read_request_bytes"] = int(round(sum(rq) / len(rq) * 128))
print(json.dumps(out, indent=1))
