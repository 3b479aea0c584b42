# round 6, call 1: the set-form selection on its own (tools/exp_select.py) against the previous build, then parity, then a per-kernel A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_sel; mkdir -p $O
for rep in 1 2; do
python tools/exp_select.py > $O/select_new_$rep.txt 2>&1; cat $O/select_new_$rep.txt
MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=$GRAFT_REPO_ROOT/quantization_amd/lib/libmcq_prev.so python tools/exp_select.py > $O/select_prev_$rep.txt 2>&1; cat $O/select_prev_$rep.txt
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
bash tools/ab_kernels.sh > /dev/null 2>&1
python tools/ab_kernels_show.py > $O/ab_kernels.txt 2>&1; cat $O/ab_kernels.txt
