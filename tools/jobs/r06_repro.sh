cd $GRAFT_REPO_ROOT
# round 6, last: the whole GPU suite three times on the final library (073c168), smoke(), and the stress once more
O=gpurun_out/r6x_suite3; mkdir -p $O
for i in 1 2 3; do (timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3); done | tee $O/suite_x3.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.txt
gcc -std=c99 -O2 -Wall -Wextra -I include -I tests/c tests/c/stream_stress.c -o /tmp/stream_stress -L go-snark-study_amd -lgosnark_hip -Wl,-rpath,$PWD/go-snark-study_amd -lpthread && timeout 600 /tmp/stream_stress 16 8 60 3 2 1 2 2>&1 | tail -8 | tee $O/stream_stress_60s.txt
timeout 600 python tools/soak_mixed.py 240 29 2>&1 | tail -4 | cut -c1-200 | tee $O/soak.txt
