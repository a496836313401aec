# Round 5, tenth GPU call: quotient with the spectrum product inside the first inverse pass -- parity (polynomial + prover tests) and
# the px-route headline, blocking and pipelined.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5j
mkdir -p gpurun_out/$T
(timeout 1200 python -m pytest tests/test_gpu_prove.py -q --maxfail=5 2>&1 | tail -6) > gpurun_out/$T/pytest_prove.txt; tail -3 gpurun_out/$T/pytest_prove.txt
bash tools/gpu_run.sh $T bench pipelined --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-extras --no-check -- bench blocking --pipeline 1 --steps 8 --warmup 2 --reps 3 --cpu-log2n 0 --no-extras --no-check
GS_NO_OVERLAP=1 bash tools/gpu_run.sh $T bench serialised --pipeline 1 --steps 6 --warmup 2 --reps 3 --cpu-log2n 0 --no-extras --no-check
