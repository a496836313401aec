#!/bin/bash
# Round 5: the command lines behind every profiles/r05_* file that is NOT part of the final measurement (r05_final_measurement.sh).
# Run one experiment per gpurun call, from the repo root of the snapshot:
#     gpurun --timeout 1500 -- 'bash tools/jobs/r05_experiments.sh <name>'
# Output goes to gpurun_out/r5x_<name>/; what is worth keeping is copied by hand into profiles/.  Experiments that need the
# development library build it first HERE (build container):  make -C go-snark-study_amd/csrc EXTRA=-DGS_DEV_KNOBS BUILD=build_dev
# LIB=../../gpurun_variants/lib_dev.so   (GS_LIB selects it; the product library has every GS_* tuning variable compiled out).
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
NAME=${1:?experiment name}; T=r5x_$NAME; OUT=gpurun_out/$T; mkdir -p $OUT
line() {   # <label> <env...> -- <bench args>: one bench.py line, summarised
  local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --cpu-log2n 0 --no-extras --no-check "$@" 2>/dev/null | tail -1 > $OUT/.line.json
  python - "$label" $OUT/.line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read()); t = d["device_ms_per_step"]
print("%-30s %8.3f ms (min %.3f) acc g1 %.2f g2 %.2f poly %.2f plan %.2f reduce %.2f heavy %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_min"],
      t["acc_g1_ms"], t["acc_g2_ms"], t["poly_ms"], t["plan_ms"], t["reduce_ms"], d["plan_per_step"]["heavy_buckets"]))
PY
}
streams() {  # <env...>: ms per proof of bench.py's five distinct-witness streams (one process)
  echo -n "$*: "; env "$@" GS_HOST_STAGE=${GS_HOST_STAGE:-1} timeout 600 python tools/stream_host_ab.py --child 20 2>$OUT/host_trace.txt | tail -1
}
case $NAME in
  host_stage)            # r05_ab_host_stage.txt: event-ordered copy / host waits / copy on the readers' streams
    timeout 900 python tools/stream_host_ab.py 20 | tee $OUT/ab_host_stage.txt ;;
  host_runtime_switches) # r05_ab_host_ticket_runtime_switches.txt
    for v in "GS_X=0" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=0" "GS_HOST_STAGE=0" "GS_HOST_STAGE=0 GPU_MAX_HW_QUEUES=8"; do streams $v; done | tee $OUT/ab.txt ;;
  stage_pieces)          # r05_ab_stage_pieces.txt: staging piece size x buffers x copy threads, host times of _begin (GS_HOST_TRACE)
    for v in "GS_STAGE_MIB=4 GS_STAGE_BUFFERS=2 GS_COPY_THREADS=4" "GS_STAGE_MIB=8" "GS_STAGE_MIB=16" "GS_STAGE_MIB=32" "GS_STAGE_MIB=8 GS_COPY_THREADS=1" "GS_STAGE_MIB=8 GS_COPY_THREADS=16"; do
      streams GS_HOST_TRACE=1 $v; grep "begin:" $OUT/host_trace.txt | sed -n '40,42p'; grep staged_h2d $OUT/host_trace.txt | sed -n '100,103p'
    done | tee $OUT/ab_stage_pieces.txt
    python tools/h2d_bandwidth.py | tee $OUT/h2d_bandwidth.txt ;;
  issue_cycles)          # r05_ubench_issue.txt, r05_ubench_mulmod_real_cycles.txt (hipcc --offload-arch=gfx950 -O3 tools/ubench_*.hip in the build container)
    ./tools/ubench_issue | tee $OUT/ubench_issue.txt; ./tools/ubench_mulmod | tee $OUT/ubench_mulmod.txt ;;
  large)                 # r05_bench_line_2p24.json, r05_bench_line_2p22*.json
    python bench.py --log2n 24 --table-policy never --steps 3 --warmup 1 --reps 2 --cpu-log2n 0 --no-extras | tail -1 > $OUT/bench_2p24.json
    for pol in always never; do python bench.py --log2n 22 --table-policy $pol --steps 4 --warmup 1 --reps 3 --cpu-log2n 0 --no-extras | tail -1 > $OUT/bench_2p22_$pol.json; done ;;
  table_free_traffic)    # r05_bench_line_table_free.json (live PMC passes of the table-free route)
    python bench.py --table-policy never --cpu-log2n 0 --no-check --steps 10 --warmup 3 --reps 3 | tail -1 > $OUT/bench_table_free.json ;;
  soak)                  # r05_soak_mixed.txt
    timeout 600 python tools/soak_mixed.py 200 11 | tail -45 | tee $OUT/soak_mixed.txt ;;
  realistic_schedule)    # r05_ab_realistic_plan_w_stream.txt, r05_timeline_realistic_*.txt
    bash tools/gpu_run.sh $T env GS_PLANW_STREAM=2 GS_PLANW_STREAM=0 GS_PLANW_STREAM=2,GS_TAIL_FLIP=0 : --instance realistic --steps 12 --warmup 3 --reps 3
    D=/tmp/prof_real; rm -rf $D
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python "$OLDPWD/bench.py" --instance realistic --steps 12 --warmup 3 --reps 1 --cpu-log2n 0 --no-extras --no-check > /dev/null 2>&1 )
    python tools/trace_steady.py "$(find $D -name '*kernel_trace.csv' | head -1)" "k_bucket_accumulate<gs::Fq2Tag>" 8 2 > $OUT/steady_realistic.txt ;;
  chunk_sweep)           # r05_sweep_chunk.txt (development library: GS_CHUNK forces the plan's chunk)
    export GS_LIB=$PWD/gpurun_variants/lib_dev.so
    for k in 8 12 16 20 24 28 32 36 40 48 64; do
      line dense_c$k GS_CHUNK=$k -- --steps 10 --warmup 2 --reps 3
      line realistic_c$k GS_CHUNK=$k -- --instance realistic --steps 10 --warmup 3 --reps 3
      line gates_c$k GS_CHUNK=$k -- --instance gates --steps 10 --warmup 2 --reps 3
      line msm_g1_c$k GS_CHUNK=$k -- --workload msm_g1 --steps 40 --warmup 5 --reps 3
      line p2p18_c$k GS_CHUNK=$k -- --log2n 18 --steps 40 --warmup 5 --reps 3
    done | tee $OUT/sweep_chunk.txt ;;
  hx_fused)              # r05_ab_hx_fused_passes.txt (development library: GS_HX_UNFUSED=1 = the separate point-wise kernels)
    export GS_LIB=$PWD/gpurun_variants/lib_dev.so
    bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --steps 12 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --pipeline 1 --steps 10 --warmup 3 --reps 3
    GS_NO_OVERLAP=1 bash tools/gpu_run.sh $T env GS_HX_UNFUSED=1 : --workload prove_witness --pipeline 1 --steps 6 --warmup 2 --reps 3 ;;
  sparse_b)              # r05_ab_sparse_b_split.txt
    bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --steps 10 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --pipeline 1 --steps 8 --warmup 2 --reps 3
    bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --workload prove_witness --steps 10 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 GS_SPLIT_B_PERCENT=75 : --instance realistic --steps 12 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --workload prove_pinocchio --log2n 18 --steps 20 --warmup 3 --reps 3 ;;
  background_builds)     # r05_background_build_slabs.txt: the warm-up transient of a fresh key under policy auto
    for lg in 18 17 16 15 14; do echo "== GS_TABLE_BG_SLAB_LOG2=$lg"; GS_TABLE_BG_SLAB_LOG2=$lg timeout 300 python tools/time_first_proof.py auto 20 40; done 2>&1 | grep -v amdgpu.ids | tee $OUT/slabs.txt
    timeout 300 python tools/time_first_proof.py auto 16 30 | tail -2 | tee -a $OUT/slabs.txt
    timeout 300 python tools/time_first_proof.py always 20 6 | tee -a $OUT/slabs.txt ;;
  sq_issue)              # r05_pmc_sq_issue_breakdown*.txt: SQ counters of the G1 accumulation, one --pmc pass per group
    bash tools/gpu_run.sh $T pmc sq_issue k_bucket_accumulate --steps 2 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check : SQ_WAVE_CYCLES SQ_BUSY_CYCLES : SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU : SQ_WAIT_INST_ANY SQ_WAIT_ANY : SQ_INSTS_VALU SQ_INSTS_SALU : SQ_ACTIVE_INST_ANY SQ_WAVES ;;
  *) echo "unknown experiment $NAME"; exit 2 ;;
esac
