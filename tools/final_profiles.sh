set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-final_r06}
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc-check > $O/bench_under_rocprof.json 2> $O/kt.err
for p in 1 2; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$p -- python $R/tools/exp_trainer_profile.py $p > $O/tr$p.log 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python $R/tools/exp_decode_ab.py MCQ_DECODE_BLK=1 > $O/dec.log 2>&1
bash $R/tools/pmc_passes.sh ${1:-final_r06}/pmc > $O/pmc.log 2>&1
cd $R
python bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --single-device --no-cpu-baseline --dp-iters 100 > $O/bench_dp2.json 2> $O/bench_dp2.err
ls $O
