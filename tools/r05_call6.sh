set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-pmc-check"
for v in 0 1; do
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  tag=$(echo $c | cut -d' ' -f1)
  MCQ_TABLE1_LEAN=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/lean${v}_$tag -- python $R/bench.py $ARGS > $O/lean${v}_$tag.log 2>&1; echo "lean$v $tag rc=$?"
done; done
python $R/tools/pmc_table.py $O k_tf_level1 > $O/level1_counters.txt 2>&1
