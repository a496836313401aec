# round 6: the suite once more on a fresh box (a run of it had one subprocess time out: tests/test_gpu_msm.py::test_randomised_degenerate_sums..., not reproducible
# standalone), then the table builder's speed with Jacobian doublings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x_repro
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --durations=15 2>&1 | tail -40) > gpurun_out/r6x_repro/pytest_gpu.txt; tail -5 gpurun_out/r6x_repro/pytest_gpu.txt
for pol in always auto; do timeout 300 python tools/time_first_proof.py $pol 20 30 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6x_repro/first_proof_jacobian_builder.txt
for l in 16 18; do timeout 300 python tools/time_first_proof.py auto $l 12 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r6x_repro/first_proof_jacobian_builder.txt
