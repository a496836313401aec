#!/bin/bash
# The ONE runner for everything that happens on the GPU box (replaces the frozen gpu_*.sh copies of rounds 1-2).
# Run it THROUGH gpurun, from the repo root of the snapshot:
#     gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <action> [args] [-- <action> [args]] ...'
# Everything is written under gpurun_out/<tag>/ (merged back into the build container); summaries worth keeping are
# copied by hand into profiles/ (tracked).  Actions (chain them with `--`; they run in order, one GPU, one box, so A/B
# numbers inside one call are comparable -- box-to-box spread is +-3 % and sometimes 10 %):
#
#   tests [pytest args]               python -m pytest tests -m gpu -x -q [args]            -> pytest.txt
#   bench <name> [bench.py args]      one bench.py line                                      -> bench_<name>.json (+ one summary line)
#   ab <variant>... [: bench args]    bench.py (--no-check --no-extras --cpu-log2n 0) for the in-tree library and every
#                                     gpurun_variants/lib_<variant>.so (GS_LIB), two rounds, pipelined + blocking 2^20 and
#                                     pipelined 2^16 unless bench args are given after ':'       -> ab.txt
#   env <NAME=VALUE[,NAME=VALUE]>... [: bench args] same, but the variants are environment settings (e.g. GS_FOLD_MAX=16)  -> ab_env.txt
#   stats <name> [bench args]         rocprofv3 --kernel-trace --stats of a bench.py run      -> stats_<name>.csv
#   trace <name> [bench args]         rocprofv3 --kernel-trace, timeline of the LAST step (start ms, duration ms, kernel)
#                                     + per-kernel sums; GS_NO_OVERLAP=1 in the environment serialises the streams -> timeline_<name>.txt
#   pmc <name> <kernel-substring> <bench args> : <counters of pass 1> [: <counters of pass 2> ...]
#                                     one rocprofv3 --pmc pass per counter group (never combined with tracing), per-dispatch
#                                     averages of the kernels whose name contains the substring  -> pmc_<name>.txt
#   run <name> <command...>           anything else, output captured                          -> run_<name>.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"

summary_line() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); t = d.get("device_ms_per_step", {})
        print("median %.3f min %.3f ms/step | value %.4g %s | acc g1 %.2f g2 %.2f poly %.2f plan %.2f reduce %.2f | valu frac %.3f" % (
            d["ms_per_step"], d.get("ms_per_step_min", 0), d["value"], d["unit"], t.get("acc_g1_ms", 0), t.get("acc_g2_ms", 0), t.get("poly_ms", 0),
            t.get("plan_ms", 0), t.get("reduce_ms", 0), d.get("roofline_valu", {}).get("frac", 0)))
PY
}
quick_bench() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 > "$OUT/.last.json"; summary_line "$OUT/.last.json"; }
default_ab() {   # $1 = label
  echo -n "2^20 pipelined, $1: "; quick_bench --steps 10 --warmup 3 --reps 5
  echo -n "2^20 blocking,  $1: "; quick_bench --steps 8 --warmup 2 --reps 3 --pipeline 1
  echo -n "2^16 pipelined, $1: "; quick_bench --log2n 16 --steps 100 --warmup 10 --reps 3
}
timeline_py() { python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
g2 = [i for i, r in enumerate(rows) if "k_bucket_accumulate<gs::Fq2Tag>" in r["Kernel_Name"]]
if g2:   # the last operation: from the k_digits before its G2 accumulation to the end
    i0 = g2[-1]
    while i0 > 0 and "k_digits" not in rows[i0]["Kernel_Name"]: i0 -= 1
    rows = rows[max(0, i0 - 8):]
else:
    rows = rows[-400:]
t0 = rows[0]["s"]; tot = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:52]
    d = (r["e"] - r["s"]) / 1e6
    tot[name] = tot.get(name, 0) + d
    print("%9.3f %8.3f  %s" % ((r["s"] - t0) / 1e6, d, name))
print("--- per kernel, summed over the window")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print("%8.3f  %s" % (v, k))
PY
}

while [ $# -gt 0 ]; do
  ACT=$1; shift
  ARGS=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  case $ACT in
    tests)
      ( timeout 1500 python -m pytest tests -m gpu -x -q "${ARGS[@]}" 2>&1 | tail -15 ) | tee "$OUT/pytest.txt" ;;
    bench)
      NAME=${ARGS[0]}; ( timeout 900 python bench.py "${ARGS[@]:1}" 2>&1 | tail -1 ) > "$OUT/bench_$NAME.json"
      echo -n "bench $NAME: "; summary_line "$OUT/bench_$NAME.json" ;;
    ab|env)
      VARS=(); BARGS=(); seen=0
      for a in "${ARGS[@]}"; do if [ "$a" = ":" ]; then seen=1; elif [ $seen = 0 ]; then VARS+=("$a"); else BARGS+=("$a"); fi; done
      F=$OUT/ab.txt; [ $ACT = env ] && F=$OUT/ab_env.txt
      for round in 1 2; do for v in "" "${VARS[@]}"; do
        ( if [ -n "$v" ]; then if [ $ACT = ab ]; then export GS_LIB=$ROOT/gpurun_variants/lib_$v.so; else for kv in ${v//,/ }; do export "$kv"; done; fi; fi
          if [ ${#BARGS[@]} -gt 0 ]; then echo -n "${BARGS[*]}, ${v:-default}: "; quick_bench "${BARGS[@]}"; else default_ab "${v:-default}"; fi )
      done; done 2>&1 | tee -a "$F" ;;
    stats)
      NAME=${ARGS[0]}; D=/tmp/prof_$NAME; rm -rf $D
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python "$ROOT/bench.py" "${ARGS[@]:1}" > "$OUT/stats_${NAME}_run.txt" 2>&1 )
      S=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" "$OUT/stats_$NAME.csv" && head -12 "$OUT/stats_$NAME.csv"
      rm -rf $D ;;
    trace)
      NAME=${ARGS[0]}; D=/tmp/prof_$NAME; rm -rf $D
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python "$ROOT/bench.py" "${ARGS[@]:1}" > "$OUT/trace_${NAME}_run.txt" 2>&1 )
      T=$(find $D -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && timeline_py "$T" > "$OUT/timeline_$NAME.txt" && tail -22 "$OUT/timeline_$NAME.txt"
      rm -rf $D ;;
    pmc)
      NAME=${ARGS[0]}; KSUB=${ARGS[1]}; BARGS=(); PMCGRP=(); cur=""; seen=0
      for a in "${ARGS[@]:2}"; do
        if [ "$a" = ":" ]; then [ $seen = 1 ] && PMCGRP+=("$cur"); cur=""; seen=1
        elif [ $seen = 0 ]; then BARGS+=("$a"); else cur="$cur $a"; fi
      done; [ -n "$cur" ] && PMCGRP+=("$cur")
      : > "$OUT/pmc_$NAME.txt"; i=0
      for GRP in "${PMCGRP[@]}"; do
        i=$((i+1)); D=/tmp/pmc_${NAME}_$i; rm -rf $D
        ( cd /tmp && timeout 900 rocprofv3 --pmc $GRP --output-format csv -d $D -o pmc -- python "$ROOT/bench.py" "${BARGS[@]}" > "$OUT/pmc_${NAME}_run$i.txt" 2>&1 )
        C=$(find $D -name "*counter_collection.csv" | head -1)
        for CN in $GRP; do [ -n "$C" ] && python tools/pmc_summary.py "$C" $CN 40 | grep -F "$KSUB" | sed "s/^/$CN: /"; done | tee -a "$OUT/pmc_$NAME.txt"
        rm -rf $D
      done ;;
    run)
      NAME=${ARGS[0]}; ( timeout 1500 "${ARGS[@]:1}" 2>&1 | tail -60 ) | tee "$OUT/run_$NAME.txt" ;;
    *) echo "gpu_run.sh: unknown action $ACT" >&2; exit 2 ;;
  esac
done
