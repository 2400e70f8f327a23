#!/bin/bash
# usage: bash tools/kernel_ab.sh <tag> <kernel substring> "<EXTRA A>" "<EXTRA B>" ...  — alone-time (single stream) of one kernel per build variant
tag=$1; shift; kern=$1; shift; mkdir -p gpurun_out/$tag; R=$PWD
for ex in "$@"; do
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 $KAB_ENV && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $KAB_ARGS > $R/gpurun_out/$tag/prof.log 2>&1)
  f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$kern" "$ex" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]) and not r["Name"].endswith("_inst"): print("[",sys.argv[3],"]", r["Name"], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3))
PY
done
rm -rf gpurun_out/$tag/prof
