#!/bin/bash
# Round 6: the command lines behind the profiles/r06_* files.  One experiment per gpurun call, from the repo root of the snapshot:
#     gpurun --timeout 1500 -- 'bash tools/jobs/r06_experiments.sh <name>'
# Output goes to gpurun_out/r6x_<name>/; what is worth keeping is copied by hand into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
NAME=${1:?experiment name}; T=r6x_$NAME; OUT=gpurun_out/$T; mkdir -p $OUT
case $NAME in
  power)                 # r06_power_clock_trace.txt, r06_ubench_issue.txt: socket power + sclk from the driver's telemetry beside the s_memtime clocks
    timeout 600 python tools/power_trace.py 4 20 2>&1 | tee $OUT/power_trace.txt
    ./tools/ubench_issue | tee $OUT/ubench_issue.txt
    ./tools/ubench_mulmod | tee $OUT/ubench_mulmod.txt
    # the clock of the bare dots3 loop from GRBM_GUI_ACTIVE (cycles the GPU was active / kernel duration), like the accumulation kernel's in r05
    D=/tmp/pmc_dots3; rm -rf $D
    ( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o p -- "$OLDPWD/tools/ubench_mulmod" 2 0 > /dev/null 2>&1 )
    python - $D <<'PY' | tee $OUT/pmc_dots3_clock.txt
import csv, glob, sys
cc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no rocprofv3 output", cc, kt); sys.exit(0)
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt[0]))}
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        d = dur.get(r["Dispatch_Id"])
        v = float(r["Counter_Value"])
        print("%s dispatch %s: GRBM_GUI_ACTIVE %.0f, duration %s ns -> %.3f GHz (counter / duration; the counter is summed over XCDs if > 8 GHz: /8 = %.3f)" % (
            r["Kernel_Name"][:40], r["Dispatch_Id"], v, d, v / d if d else 0, v / d / 8 if d else 0))
PY
    ;;
  cu_mask)               # r06_ubench_issue_cu_mask.txt: the same loops on 8 / 32 / 256 CUs -- is an instruction's wall-time price a chip-wide (power / current) effect?
    for cus in 8 32 128 0; do
      for mode in 11 7 0 9 18; do ./tools/ubench_issue $mode 1.5 $cus | grep -v "^gfx"; done
    done 2>&1 | tee $OUT/ubench_issue_cu_mask.txt ;;
  placement)             # r06_ubench_placement.txt: where and when the waves of the ubench launches run
    ./tools/ubench_placement 2>&1 | tee $OUT/ubench_placement.txt ;;
  mask_placement)        # r06_ubench_placement_cu_mask.txt: do CU-masked streams confine the waves (HW_ID / XCC_ID of every wave)?
    ./tools/ubench_placement masked 2>&1 | tee $OUT/ubench_placement_cu_mask.txt
    ./tools/ubench_placement 2>&1 | tee $OUT/ubench_placement.txt ;;
  auto_transient)        # r06_auto_instalments.txt: every blocking proof of a fresh 2^20 key under `auto` at several build budgets, `always` beside it
    for pct in 100 50 200 400 100000; do echo "== GS_TABLE_BUDGET_PCT=$pct"; GS_TABLE_BUDGET_PCT=$pct timeout 300 python tools/time_first_proof.py auto 20 40; done 2>&1 | grep -v "^$" | tee $OUT/auto_instalments.txt
    echo "== policy always"; timeout 300 python tools/time_first_proof.py always 20 12 2>&1 | tee -a $OUT/auto_instalments.txt
    echo "== auto, 2^16 and 2^18"; for l in 16 18; do timeout 300 python tools/time_first_proof.py auto $l 40; done 2>&1 | tee -a $OUT/auto_instalments.txt ;;
  c_producer)            # r06_c_producer.txt: the C ABI's ingest ceiling (tests/c/stream_producer.c at 2^20, 8 witnesses) beside bench.py's Python-driven streams
    gcc -std=c99 -O2 -Wall -Wextra -I include -I tests/c tests/c/stream_producer.c -o /tmp/stream_producer -L go-snark-study_amd -lgosnark_hip -Wl,-rpath,$PWD/go-snark-study_amd -lpthread || exit 1
    timeout 900 /tmp/stream_producer 20 8 4 1 2>&1 | tee $OUT/c_producer.txt
    gcc -O2 -pthread tools/pack_cost.c -o /tmp/pack_cost && /tmp/pack_cost 20 2>&1 | tee $OUT/pack_cost.txt
    timeout 900 python bench.py --steps 10 --warmup 3 --reps 3 --cpu-log2n 0 --no-check 2>/dev/null | tail -1 > $OUT/bench_line.json
    python - $OUT/bench_line.json <<'PY' | tee -a $OUT/c_producer.txt
import json, sys
d = json.loads(open(sys.argv[1]).read())
s = d.get("extras", d).get("stream_distinct_host", d.get("stream_distinct_host"))
print("bench.py (Python-driven, same box, same call):", json.dumps({k: (v if not isinstance(v, dict) else {"ms_per_proof": v.get("ms_per_proof"), "reps": v.get("ms_per_proof_reps")}) for k, v in (s or {}).items()}))
print("bench.py headline: %.3f ms per proof" % d["ms_per_step"])
PY
    ;;
  suite)                 # the GPU suite + smoke
    ( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) | tee $OUT/pytest.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt ;;
  quick)                 # the tests of what round 6 touched
    ( timeout 1500 python -m pytest tests/test_gpu_table_policy.py tests/test_gpu_c_drivers.py tests/test_gpu_stream_host.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 ) | tee $OUT/pytest.txt ;;
  slot_streams)          # r06_ab_slot_streams.txt: every pipelined proof's side work on its own slot's stream (GS_SLOT_STREAMS=1) vs the shared plan / polynomial stream
    for inst in sqchain realistic gates; do for wl in prove prove_witness; do
      bash tools/gpu_run.sh $T env GS_SLOT_STREAMS=1 : --instance $inst --workload $wl --steps 12 --warmup 3 --reps 3
    done; done
    bash tools/gpu_run.sh $T env GS_SLOT_STREAMS=1 : --log2n 16 --steps 100 --warmup 10 --reps 3
    bash tools/gpu_run.sh $T env GS_SLOT_STREAMS=1 : --log2n 18 --steps 40 --warmup 5 --reps 3
    bash tools/gpu_run.sh $T env GS_SLOT_STREAMS=1 : --log2n 22 --steps 4 --warmup 1 --reps 3
    bash tools/gpu_run.sh $T env GS_SLOT_STREAMS=1 : --workload prove_pinocchio --instance gates --steps 8 --warmup 2 --reps 3 ;;
  add_latency)           # r06_timeline_msm_2p16_critical_path.txt: what one dependent addition costs a lone wave + the blocking 2^16 MSM's timeline
    ./tools/ubench_add_latency 2>&1 | tee $OUT/add_latency.txt
    bash tools/gpu_run.sh $T trace msm_2p16_blocking --workload msm_g1 --log2n 16 --pipeline 1 --steps 20 --warmup 5 --reps 1 --cpu-log2n 0 --no-extras --no-check
    bash tools/gpu_run.sh $T trace prove_2p16_pipelined --log2n 16 --steps 20 --warmup 5 --reps 1 --cpu-log2n 0 --no-extras --no-check ;;
  heavy_sharing)         # r06_ab_heavy_kernels_sharing.txt: the two heavy-bucket kernels always in their sharing form (default) vs one wave per SIMD below 2^19 terms (lib_heavyalone.so = the library before the change)
    bash tools/gpu_run.sh $T ab heavyalone : --log2n 16 --steps 100 --warmup 10 --reps 5
    bash tools/gpu_run.sh $T ab heavyalone : --log2n 17 --steps 60 --warmup 10 --reps 5
    bash tools/gpu_run.sh $T ab heavyalone : --log2n 18 --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab heavyalone : --log2n 18 --instance realistic --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab heavyalone : --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 5
    bash tools/gpu_run.sh $T ab heavyalone : --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 5 ;;
  proof_clock)           # r06_power_clock_streams.txt: which part of a proof pulls the PLL down (the MSM streams run at 2.30-2.34 GHz, the proof stream at 2.03-2.08)
    timeout 600 python tools/power_trace.py 4 20 --streams-only 2>&1 | grep -v "^        \|^    \|^GPU\|^\$\|^====\|amdgpu.ids" | tee $OUT/streams_default.txt
    GS_NO_OVERLAP=1 timeout 600 python tools/power_trace.py 4 20 --streams-only 2>&1 | grep "stream\|proofs\|idle" | tee $OUT/streams_no_overlap.txt ;;
  box_spread)            # r06_box_spread.txt: the same library takes 8.5-9.2 ms per proof from box to box -- the bench line beside the box's power / sclk under the streams
    (timeout 600 python bench.py --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-check --no-extras 2>/dev/null | tail -1) > $OUT/bench_line.json
    python -c "import json,sys; d=json.load(open('$OUT/bench_line.json')); print('bench line: %.3f ms per proof, G1 accumulation launch %.3f ms, valu frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_valu']['frac']))" | tee $OUT/box_spread.txt
    timeout 600 python tools/power_trace.py 4 20 --streams-only 2>&1 | grep "^idle\|^G1 MSM stream\|^G2 MSM stream\|^Groth16 proof stream, \|^gs_r1cs_px" | tee -a $OUT/box_spread.txt
    ./tools/ubench_issue 0 2 0 | grep -v "^gfx" | tee -a $OUT/box_spread.txt
    ./tools/ubench_mulmod 2 2 2>&1 | tail -2 | tee -a $OUT/box_spread.txt ;;
  acc_waves)             # r06_ab_accumulate_fourth_wave.txt: k_bucket_accumulate<G1> capped at 128 VGPRs = four waves per SIMD (w4: 168 B of scratch per lane; w4np: without the
                         # register prefetch of the next point, 92 B) against the shipped 160 VGPRs = three.  make EXTRA="-DGS_G1_WAVES=4 [-DGS_G1_PREFETCH=0]" BUILD=build_w4[np] LIB=../../gpurun_variants/lib_w4[np].so
    bash tools/gpu_run.sh $T ab w4np w4 : --steps 10 --warmup 3 --reps 5
    bash tools/gpu_run.sh $T ab w4np w4 : --workload msm_g1 --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab w4np w4 : --log2n 18 --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab w4np w4 : --log2n 22 --steps 4 --warmup 1 --reps 3
    bash tools/gpu_run.sh $T ab w4np w4 : --instance realistic --steps 12 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T ab w4np w4 : --workload prove_pinocchio --steps 8 --warmup 2 --reps 3 ;;
  four_in_flight)        # r06_ab_four_in_flight.txt: four ticket slots instead of three (make EXTRA=-DGS_MAX_IN_FLIGHT=4 LIB=../../gpurun_variants/lib_inflight4.so), bench --pipeline 4
    line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('device_ms_per_step',{}); print('%s: median %.3f min %.3f ms/step | value %.4g %s | in flight %s | acc g1 %.2f g2 %.2f' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_min'], d['value'], d['unit'], d['config'].get('proofs_in_flight'), t.get('acc_g1_ms',0), t.get('acc_g2_ms',0)))" "$1"; }
    for args in "--steps 12 --warmup 4 --reps 5" "--workload msm_g1 --steps 40 --warmup 5 --reps 5" "--log2n 18 --steps 40 --warmup 5 --reps 5" "--log2n 16 --steps 100 --warmup 10 --reps 5" "--instance realistic --steps 12 --warmup 4 --reps 3" "--workload prove_pinocchio --steps 8 --warmup 4 --reps 3" "--workload prove_witness --steps 12 --warmup 4 --reps 3"; do
      for round in 1 2; do
        timeout 600 python bench.py $args --pipeline 3 --cpu-log2n 0 --no-extras --no-check 2>/dev/null | line "$args, shipped library, 3 in flight"
        GS_LIB=$PWD/gpurun_variants/lib_inflight4.so timeout 600 python bench.py $args --pipeline 4 --cpu-log2n 0 --no-extras --no-check 2>/dev/null | line "$args, four slots, 4 in flight"
        GS_LIB=$PWD/gpurun_variants/lib_inflight4.so timeout 600 python bench.py $args --pipeline 3 --cpu-log2n 0 --no-extras --no-check 2>/dev/null | line "$args, four slots, 3 in flight"
      done
    done 2>&1 | tee $OUT/ab.txt ;;
  proof_kernel_clocks)   # r06_pmc_kernel_clocks.txt: GRBM_GUI_ACTIVE / duration per kernel (= the clock the kernel ran at; PMC passes serialise the kernels) in a proof
                         # stream and in a G1 MSM stream -- which kernels of a proof run at the low clock the hwmon trace sees?
    for wl in "prove" "msm_g1" "prove_pinocchio"; do
      D=/tmp/pmc_clk_$wl; rm -rf $D
      ( cd /tmp && timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o p -- python $OLDPWD/bench.py --workload $wl --steps 8 --warmup 3 --reps 1 --cpu-log2n 0 --no-extras --no-check > /dev/null 2>&1 )
      echo "== workload $wl"
      python - $D <<'PY'
import csv, glob, sys, collections
cc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no rocprofv3 output", cc, kt); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]))
rows = []
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    d, t0 = dur[r["Dispatch_Id"]]
    rows.append((t0, r["Kernel_Name"], float(r["Counter_Value"]), d))
rows.sort()
rows = rows[len(rows) // 3:]                      # the steady part of the run
agg = collections.defaultdict(lambda: [0.0, 0, 0])
for _, name, v, d in rows:
    a = agg[name.split("(")[0][:70]]
    a[0] += v; a[1] += d; a[2] += 1
print("%-72s %8s %10s %9s" % ("kernel (steady two thirds of the run)", "launches", "avg us", "GHz"))
for name, (v, d, k) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    ghz = v / d
    print("%-72s %8d %10.1f %9.3f%s" % (name, k, d / k / 1e3, ghz / 8 if ghz > 4 else ghz, " (counter summed over 8 XCDs: / 8)" if ghz > 4 else ""))
PY
    done 2>&1 | tee $OUT/pmc_kernel_clocks.txt ;;
  clock_timeline)        # r06_clock_timeline.txt: the shader clock at 20 us resolution beside each of the library's streams (tools/clock_probe.hip: one wave, s_memtime against s_memrealtime)
    timeout 600 python tools/clock_timeline.py 20 60 20 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_timeline.txt ;;
  acc_block)             # r06_ab_accumulate_block.txt: 64- / 128-thread workgroups for the accumulation kernels (no LDS, no barrier: the group is only the dispatcher's unit)
    bash tools/gpu_run.sh $T ab acc64 acc128 : --steps 10 --warmup 3 --reps 5
    bash tools/gpu_run.sh $T ab acc64 acc128 : --workload msm_g1 --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab acc64 acc128 : --log2n 16 --steps 100 --warmup 10 --reps 5
    bash tools/gpu_run.sh $T ab acc64 acc128 : --log2n 18 --steps 40 --warmup 5 --reps 5
    bash tools/gpu_run.sh $T ab acc64 acc128 : --log2n 22 --steps 4 --warmup 1 --reps 3
    bash tools/gpu_run.sh $T ab acc64 acc128 : --instance realistic --steps 12 --warmup 3 --reps 3
    bash tools/gpu_run.sh $T ab acc64 acc128 : --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 5 ;;
  split_jobs)            # (in r06_power_clock_streams.txt) one accumulation launch per base array (GS_ACC_SPLIT_JOBS=1, a knob that existed for this experiment only) vs one launch with grid.y = jobs
    bash tools/gpu_run.sh $T env GS_ACC_SPLIT_JOBS=1 : --steps 10 --warmup 3 --reps 5
    bash tools/gpu_run.sh $T env GS_ACC_SPLIT_JOBS=1 : --workload prove_pinocchio --steps 8 --warmup 2 --reps 3
    bash tools/gpu_run.sh $T env GS_ACC_SPLIT_JOBS=1 : --log2n 18 --steps 40 --warmup 5 --reps 5
    GS_ACC_SPLIT_JOBS=1 timeout 600 python tools/power_trace.py 4 20 --streams-only 2>&1 | grep "stream\|proofs\|idle" | tee $OUT/streams_split_jobs.txt ;;
  *) echo "unknown experiment $NAME" >&2; exit 2 ;;
esac
