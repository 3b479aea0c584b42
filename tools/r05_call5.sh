set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5
mkdir -p $O
cd $R
for rep in 1 2; do for v in 0 1; do
MCQ_TABLE1_LEAN=$v python tools/exp_profile_shapes.py 1024,16,256,65536 512,8,256,4096 512,8,256,65536 256,4,256,65536 > $O/shapes_lean${v}_$rep.txt 2>&1
MCQ_TABLE1_LEAN=$v python tools/ab_trainer.py > $O/trainer_lean${v}_$rep.txt 2>&1
done; done
grep -h "encode\|step" $O/*.txt | head -60
