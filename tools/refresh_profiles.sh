#!/bin/bash
# copy the summaries of one tools/final_profiles.sh run (gpurun_out/<dir>) into profiles/ as the round's set
# usage: tools/refresh_profiles.sh gpurun_out/final_r03 [r03]
set -eu
S=$1; R=${2:-r06}; P=$(dirname $0)/../profiles
grep "^{" $S/bench.json | tail -1 > $P/${R}_bench.json
grep "^{" $S/bench_under_rocprof.json | tail -1 > $P/${R}_bench_under_rocprof.json
grep "^{" $S/bench_dp2.json | tail -1 > $P/${R}_bench_dp2_dryrun.json
cp $(ls -t $S/kt/*/*kernel_stats.csv | head -1) $P/${R}_kernel_stats.csv
cp $(ls -t $S/tr1/*/*kernel_stats.csv | head -1) $P/${R}_trainer_phase1_kernel_stats.csv
cp $(ls -t $S/tr2/*/*kernel_stats.csv | head -1) $P/${R}_trainer_phase2_kernel_stats.csv
if ls $S/dec/*/*kernel_stats.csv >/dev/null 2>&1; then cp $(ls -t $S/dec/*/*kernel_stats.csv | head -1) $P/${R}_decode_kernel_stats.csv; grep '^D=' $S/dec.log > $P/${R}_decode_shapes.txt; fi
python $(dirname $0)/pmc_table.py $S/pmc > $P/${R}_pmc_counters.txt
python $(dirname $0)/pmc_traffic.py $S/pmc > $P/${R}_pmc_traffic.json
ls -la $P | grep ${R}_
