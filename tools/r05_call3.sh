set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-pmc-check"
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_DATA_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/now_$tag -- python $R/bench.py $ARGS > $O/now_$tag.log 2>&1; echo "now $tag rc=$?"
  MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=$R/quantization_amd/lib/libmcq_prev.so rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prev_$tag -- python $R/bench.py $ARGS > $O/prev_$tag.log 2>&1; echo "prev $tag rc=$?"
done
rocprofv3 -L 2>/dev/null | grep -i "TCP_\|TA_" | head -80 > $O/counters_list.txt
