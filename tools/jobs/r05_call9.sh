# Round 5, ninth GPU call: the H-values stage with k_hx_weigh / k_pw_mul_bcast fused into the neighbouring NTT passes -- parity (every
# witness-route test) and an A/B against the separate kernels (development build: GS_HX_UNFUSED=1), pipelined / blocking / serialised.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5i
mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_stream_host.py -q -k "witness or values or eval or host or realistic or fall or pinocchio" 2>&1 | tail -8) > gpurun_out/$T/pytest_witness.txt; tail -3 gpurun_out/$T/pytest_witness.txt
export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --steps 12 --warmup 3 --reps 3
bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --pipeline 1 --steps 10 --warmup 3 --reps 3
GS_NO_OVERLAP=1 bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --pipeline 1 --steps 6 --warmup 2 --reps 3
