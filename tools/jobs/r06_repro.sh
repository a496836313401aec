cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x_repro
timeout 900 python tools/soak_mixed.py 420 21 2>&1 | tail -45 | tee gpurun_out/r6x_repro/soak_mixed_final.txt
timeout 300 python tools/stress_msm_random.py 120 23 2>&1 | tail -3 | tee gpurun_out/r6x_repro/stress_msm_random.txt
gcc -std=c99 -O2 -Wall -Wextra -I include -I tests/c tests/c/stream_stress.c -o /tmp/stream_stress -L go-snark-study_amd -lgosnark_hip -Wl,-rpath,$PWD/go-snark-study_amd -lpthread && timeout 600 /tmp/stream_stress 16 8 60 3 2 1 2 2>&1 | tail -8 | tee gpurun_out/r6x_repro/stream_stress_60s.txt
