cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fgemm4; mkdir -p $O
MCQ_FGEMM4=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 200 -k "logits or fixture or config or large or full_size or ragged" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/parity.log
MCQ_FGEMM4=1 timeout 600 python tools/fuzz_shapes.py 30 5 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
bash tools/ab_env_kernels.sh MCQ_FGEMM4 0 1 2>&1 | tee $O/ab.txt
