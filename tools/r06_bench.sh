cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_bench; mkdir -p $O
(time python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"; tail -c 600 $O/bench.err
python bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --single-device --no-cpu-baseline --dp-iters 100 > $O/bench_dp2.json 2> $O/bench_dp2.err; echo "dp2 rc=$?"; tail -c 400 $O/bench_dp2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench/bench.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step')})
print(json.dumps(d['roofline'])[:1500])
print(json.dumps(d['cpu_baseline'])[:1200])
print(json.dumps(d['parity'])[:800])
print(json.dumps(d['trainer_step'])[:1500])
print({k:v['ms_per_encode'] for k,v in d['kernels'].items()})
d2=json.loads(open('gpurun_out/r06_bench/bench_dp2.json').read().strip().split('\n')[-1])
print(json.dumps(d2.get('configs'))[:800]); print(json.dumps(d2.get('dp_trainer'))[:1500])
PY
