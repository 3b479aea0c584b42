cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pass16_n8; mkdir -p $O
AB_TAG=pass16 timeout 900 python tools/fuzz_pass16.py 24 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -16 $O/fuzz.log
AB_TAG=separate MCQ_PASS16=0 timeout 600 python tools/fuzz_pass16.py 0 > $O/separate.log 2>&1; tail -6 $O/separate.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "b4_p1 or b8_p1 or k16_n or outlier300" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
timeout 1200 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_trainer_config_e.py -q -x --timeout 300 > $O/trainer.log 2>&1; echo "trainer rc=$?"; tail -5 $O/trainer.log
