set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c7
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_dp2.json 2> $O/bench_dp2.err; echo "bench --gpus 2 (self-launched, gloo, one device) rc=$?"
head -c 600 $O/bench_dp2.json
