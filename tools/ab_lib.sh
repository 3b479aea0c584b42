#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <tag> ; libmcq_alt.so beside libmcq_hip.so
for rep in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-secondary > gpurun_out/ab_main_$rep.json 2>/dev/null
MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=$GRAFT_REPO_ROOT/quantization_amd/lib/libmcq_alt.so python bench.py --steps 10 --warmup 3 --no-secondary > gpurun_out/ab_alt_$rep.json 2>/dev/null
done
