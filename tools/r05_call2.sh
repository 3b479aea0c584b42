set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.log 2>&1; echo "parity rc=$?"
tail -3 $O/parity.log
bash tools/ab_kernels.sh > $O/ab.log 2>&1
python bench.py --no-pmc-check --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
head -c 400 $O/bench.json
