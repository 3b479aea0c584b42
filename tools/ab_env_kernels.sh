#!/bin/bash
# same-box per-kernel A/B over the values of one environment variable: tools/ab_env_kernels.sh VAR v1 v2 [bench args...]
# (rocprofv3 --kernel-trace --stats of bench.py --no-secondary, three interleaved runs per value; prints the k_tf / k_fgemm averages)
VAR=$1; A=$2; B=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2 3; do for v in $A $B; do
env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abe_${v}_$r -- python $R/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-profile --no-pmc-check "$@" > /dev/null 2>&1
done; done
python - "$A" "$B" <<'PY'
import csv, glob, os, sys, statistics
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
def load(d):
    f = glob.glob(f"{root}/{d}/*/*kernel_stats.csv")[0]
    return {r["Name"].split("(")[0].replace("void mcq::", "").replace("mcq::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if "mcq" in r["Name"]}
vals = sys.argv[1:3]
runs = {v: [load(f"abe_{v}_{r}") for r in (1, 2, 3)] for v in vals}
names = sorted(set().union(*[set(x) for v in vals for x in runs[v]]))
for k in names:
    if not ("k_tf" in k or "fgemm" in k or "fix_rows<" in k):
        continue
    row = []
    for v in vals:
        xs = [x.get(k, float("nan")) for x in runs[v]]
        row.append("%s: %s" % (v, " ".join("%7.1f" % x for x in xs)))
    print("%-34s %s" % (k[:34], "   ".join(row)))
PY
