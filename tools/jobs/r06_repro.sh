cd $GRAFT_REPO_ROOT
# round 6, late: is the slow first / second proof of a fresh key under `auto` (52 / 208 ms in one bench call instead of 23 / 18) reproducible?
O=gpurun_out/r6x_cold; mkdir -p $O
for i in 1; do echo "== fresh process $i"; timeout 300 python tools/time_first_proof.py auto 20 6 2>&1 | grep -v amdgpu.ids | tail -4; done | tee $O/first_proofs.txt
for i in 1 2 3 4 5; do timeout 600 python bench.py --steps 10 --warmup 3 --reps 3 --cpu-log2n 0 --no-check 2>/dev/null | tail -1 > $O/bench_line_$i.json; python - $O/bench_line_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
a = d["cold"]["auto"]
print("bench", round(d["ms_per_step"], 3), "auto", a["proofs_ms"][:4], a["time_to_steady_ms"], a["slowest_proof_after_the_first_over_steady"], a["which_run"][:8], "other", a["other_run"]["proofs_ms"][:4], a["other_run"]["time_to_steady_ms"], "always", d["cold"]["always"]["first_proof_ms"])
PY
done | tee $O/bench_cold_summary.txt
