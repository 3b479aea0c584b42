#!/bin/bash
# rocprofv3 evidence for ONE named shape: kernel-trace stats of its encode (bench.py --no-secondary: every
# launch has that shape), of its decode, and the --pmc passes the bench line's per-config roofline reads.
# usage: tools/profile_config.sh <tag> <dim> <num_codebooks>      ->  gpurun_out/<tag>/{kt,dec,pmc}
set -u
R=$GRAFT_REPO_ROOT
TAG=$1; DIM=$2; NCB=$3
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--dim $DIM --num-codebooks $NCB --no-cpu-baseline --no-secondary --no-pmc-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 5 --warmup 2 $ARGS > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python $R/tools/exp_decode_shape.py $DIM $NCB 65536 > $O/dec.log 2>&1
for pass in "tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -- python $R/bench.py --steps 1 --warmup 0 --no-profile $ARGS > $O/pmc_$name.log 2>&1
  echo "pass $name rc=$?"
done
python $R/tools/pmc_table.py $O/pmc > $O/pmc_counters.txt 2>&1
python $R/tools/pmc_traffic.py $O/pmc $DIM $NCB > $O/pmc_traffic.json 2> $O/pmc_traffic.err
find $O -name "*kernel_stats.csv" | head
