cd $GRAFT_REPO_ROOT
O=gpurun_out/long_fuzz; mkdir -p $O
timeout 1500 python tools/fuzz_shapes.py 400 11 > $O/shapes.log 2>&1; echo "shapes rc=$?"; tail -1 $O/shapes.log
timeout 600 python tools/fuzz_pass16.py 200 3 > $O/pass16.log 2>&1; echo "pass16 rc=$?"; grep "^cases" $O/pass16.log
timeout 600 python tools/fuzz_decode.py 200 7 > $O/decode.log 2>&1; echo "decode rc=$?"; tail -1 $O/decode.log
timeout 600 python tools/soak.py > $O/soak.log 2>&1; echo "soak rc=$?"; tail -2 $O/soak.log
