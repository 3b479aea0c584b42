set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_trainer_long.py -x -q -s > $O/long.log 2>&1; echo "long rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "selection or stress_mean10_d512" > $O/parity_subset.log 2>&1; echo "parity rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/profile_config.sh c1/cfgD 1024 16 > $O/cfgD.log 2>&1
bash tools/profile_config.sh c1/cfgA 256 4 > $O/cfgA.log 2>&1
tail -5 $O/long.log; tail -3 $O/parity_subset.log; head -c 1500 $O/bench.json
