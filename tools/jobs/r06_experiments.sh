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
  *) echo "unknown experiment $NAME" >&2; exit 2 ;;
esac
