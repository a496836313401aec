set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
for L in 14 16 17 18; do bash tools/gpu_run.sh r4m env GS_LIB=$DEV GS_LIB=$DEV,GS_CHUNK_H=8 GS_LIB=$DEV,GS_CHUNK_H=12 GS_LIB=$DEV,GS_CHUNK=8 GS_LIB=$DEV,GS_CHUNK=12 : --log2n $L --steps 60 --warmup 10 --reps 3; done
bash tools/gpu_run.sh r4m env GS_LIB=$DEV GS_LIB=$DEV,GS_CHUNK=8 GS_LIB=$DEV,GS_CHUNK=12 : --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
bash tools/gpu_run.sh r4m env GS_LIB=$DEV GS_LIB=$DEV,GS_CHUNK=8 GS_LIB=$DEV,GS_CHUNK=12 : --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 3
