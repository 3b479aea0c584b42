set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c8
mkdir -p $O
cd $R
MCQ_PAIR0_LOOP=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fixture or config or codes" > $O/parity_loop.log 2>&1; echo "parity(loop) rc=$?"; tail -2 $O/parity_loop.log
for rep in 1 2; do for v in 0 1 16 24 32; do
MCQ_PAIR0_LOOP=$v python tools/exp_profile_shapes.py 512,8,256,65536 2>&1 | grep -E "encode|combine_level0" | tr '\n' ' '; echo " [loop=$v]"
done; done
