cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pair0; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "selection or codes_bit_exact or full_size" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
bash tools/ab_env_kernels.sh MCQ_PAIR0_LANE 0 1 > $O/ab_pair0.txt 2>&1; cat $O/ab_pair0.txt
