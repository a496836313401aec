set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/gpurun_variants/lib_r3tails.so
DEV=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
bash tools/gpu_run.sh r4d tests
bash tools/gpu_run.sh r4d env GS_LIB=$OLD
tools/snop/run > gpurun_out/r4d/snop.txt 2>&1; cat gpurun_out/r4d/snop.txt
bash tools/gpu_run.sh r4d env GS_LIB=$DEV GS_LIB=$DEV,GS_CHUNK_H=24 GS_LIB=$DEV,GS_CHUNK_H=28 GS_LIB=$DEV,GS_CHUNK_H=36 GS_LIB=$DEV,GS_CHUNK_H=40 GS_LIB=$DEV,GS_CHUNK_H=48 : --steps 10 --warmup 3 --reps 5
for a in "10" "14 --rows 16" "16 --rows 12" "18 --rows 8"; do python tools/derive_eval_basis.py --log2n $a 2>&1 | tail -1; done | tee gpurun_out/r4d/derive_eval.txt
