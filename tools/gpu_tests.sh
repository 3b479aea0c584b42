# the GPU suite with a per-test timeout (a kernel that spins must not take the box with it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/gpu_tests; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --timeout 300 "$@" > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gpu_tests.log
