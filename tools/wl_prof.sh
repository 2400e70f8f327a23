#!/bin/bash
# usage (GPU box): bash tools/wl_prof.sh <workload> [bench args] — per-kernel ALONE averages (single stream) of a bench.py workload under rocprofv3 --kernel-trace --stats, fast mode
R=$PWD; wl=$1; shift; mkdir -p gpurun_out/wl_prof; rm -rf gpurun_out/wl_prof/p
(cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wl_prof/p -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse "$@" > $R/gpurun_out/wl_prof/log.txt 2>&1)
f=$(find gpurun_out/wl_prof/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$wl" <<'PY'
import csv, sys
print("##", sys.argv[2], "alone (single stream)")
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:18]:
    print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} pct {r["Percentage"]}')
PY
rm -rf gpurun_out/wl_prof/p
