# same-box per-kernel A/B: this build against quantization_amd/lib/libmcq_prev.so, rocprofv3 --kernel-trace --stats on each, twice, interleaved;
# python tools/ab_kernels_show.py prints the averages (how the round-4 changes to the selection and the launches were accepted or rejected)
cd /tmp && export TMPDIR=/tmp
for r in 1 2 3 4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abk_now$r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=$GRAFT_REPO_ROOT/quantization_amd/lib/libmcq_prev.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abk_prev$r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
done
echo done
