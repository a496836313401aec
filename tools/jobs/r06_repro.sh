cd $GRAFT_REPO_ROOT
# round 6, late: the context's lock made first come, first served (runtime.h FairMutex) -- the eight-thread stress at two sizes, the C producers,
# and where the waves of a CU-masked stream really run
O=gpurun_out/r6x_fair; mkdir -p $O
gcc -std=c99 -O2 -Wall -Wextra -I include -I tests/c tests/c/stream_stress.c -o /tmp/stream_stress -L go-snark-study_amd -lgosnark_hip -Wl,-rpath,$PWD/go-snark-study_amd -lpthread || exit 1
for args in "11 6 2 3 2 1 2" "16 8 20 3 2 1 2" "20 4 20 3 2 1 2"; do echo "== stream_stress $args"; timeout 600 /tmp/stream_stress $args 2>&1 | tail -8; done | tee $O/stream_stress.txt
gcc -std=c99 -O2 -Wall -Wextra -I include -I tests/c tests/c/stream_producer.c -o /tmp/stream_producer -L go-snark-study_amd -lgosnark_hip -Wl,-rpath,$PWD/go-snark-study_amd -lpthread || exit 1
timeout 900 /tmp/stream_producer 20 8 4 1 2>&1 | tee $O/c_producer.txt
./tools/ubench_placement masked 2>&1 | tee $O/ubench_placement_cu_mask.txt
timeout 900 python -m pytest tests/test_gpu_c_drivers.py tests/test_gpu_stream_host.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_c_drivers.txt
timeout 600 python bench.py --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-check 2>/dev/null | tail -1 > $O/bench_line.json
python - $O/bench_line.json <<'PY' | tee $O/bench_summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("ms_per_step", d["ms_per_step"], "value", d["value"], "build", d.get("build"))
PY
