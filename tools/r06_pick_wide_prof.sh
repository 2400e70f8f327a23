#!/bin/bash
# usage (GPU box): bash tools/r06_pick_wide_prof.sh OUT — the candidate pick of C3 (1 026 lights) ALONE (LUMEN_MI_SINGLE_STREAM=1; rocprofv3 --kernel-trace --stats), fast and
# exact mode, with the four-tile block around one LDS light table (LUMEN_MI_PICK_WIDE=1, default) and with the global-gather kernel (0); C2's pick beside it for the comparison
# VERDICT r5 item 6 asks for (profiles/r06_pick_wide.txt)
out=$1; : > $out; R=$PWD
run() {  # label, workload, mode, wide
  rm -rf gpurun_out/pw_prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 LUMEN_MI_PICK_WIDE=$4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pw_prof -- python3 $R/bench.py --workload $2 --mode $3 --steps 2 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse > $R/gpurun_out/pw_prof.log 2>&1)
  f=$(find gpurun_out/pw_prof -name "*kernel_stats.csv" | head -1)
  echo "## $1" >> $out
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pick_primary" in r["Name"] and not r["Name"].endswith("_inst") and int(r["Calls"]) > 2:
        print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
PY
  rm -rf gpurun_out/pw_prof
}
run "c2 fast (2 lights, 24-KB table, 256-thread blocks)" c2 fast 1
run "c2 exact" c2 exact 1
run "c3 fast, global gather" c3 fast 0
run "c3 fast, four tiles per block around one table" c3 fast 1
run "c3 exact, global gather" c3 exact 0
run "c3 exact, four tiles per block around one table" c3 exact 1
