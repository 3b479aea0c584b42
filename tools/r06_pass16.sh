cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pass16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "b8_p1 or k16_n" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_trainer_config_e.py -q -x --timeout 300 > $O/trainer.log 2>&1; echo "trainer rc=$?"; tail -5 $O/trainer.log
for r in 1 2; do
AB_TAG=pass16 python tools/ab_trainer_env.py 2>/dev/null
AB_TAG=separate MCQ_PASS16=0 python tools/ab_trainer_env.py 2>/dev/null
done
