set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c10
mkdir -p $O
cd $R
for m in 2 4; do
MCQ_PAIR0_MULTI=$m timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fixture or config or codes" > $O/parity_multi$m.log 2>&1; echo "parity(multi=$m) rc=$?"; tail -1 $O/parity_multi$m.log
done
for rep in 1 2 3; do for v in 0 2 4; do
MCQ_PAIR0_MULTI=$v python tools/exp_profile_shapes.py 512,8,256,65536 2>&1 | grep -E "encode|combine_level0" | tr '\n' ' '; echo " [multi=$v]"
done; done
for v in 0 2 4; do
MCQ_PAIR0_MULTI=$v python tools/exp_profile_shapes.py 1024,16,256,65536 512,8,256,4096 2>&1 | grep -E "encode|combine_level0" | tr '\n' ' '; echo " [multi=$v]"
done
