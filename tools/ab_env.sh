#!/bin/bash
# same-box A/B over the values of one environment variable: tools/ab_env.sh VAR v1 v2 ...   (bench.py --no-secondary, 2 rounds)
VAR=$1; shift
rm -f gpurun_out/ab_*.json
for rep in 1 2; do for v in "$@"; do
env $VAR=$v python bench.py --steps 10 --warmup 3 --no-secondary > gpurun_out/ab_${VAR}_${v}_$rep.json 2>/dev/null
done; done
python tools/ab_show.py
