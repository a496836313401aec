# Round 5, eighth GPU call: realistic witnesses are not accumulation-bound -- the chain plan(w) | H | plan(h) on aux 1 is as long as the
# proof period.  Does plan(w) on its own stream (GS_PLANW_STREAM=2) shorten the period?  And tail-stream variants.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5h
mkdir -p gpurun_out/$T
bash tools/gpu_run.sh $T env GS_PLANW_STREAM=2 GS_PLANW_STREAM=0 GS_PLANW_STREAM=2,GS_TAIL_FLIP=0 : --instance realistic --steps 12 --warmup 3 --reps 3
bash tools/gpu_run.sh $T env GS_PLANW_STREAM=2 : --instance realistic --workload prove_witness --steps 12 --warmup 3 --reps 3
