cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_sel2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "selection or codes_bit_exact or full_size" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
python tools/exp_select.py 2>/dev/null
python tools/exp_profile_shapes.py 1024,16,256,65536 2>/dev/null | tail -25
