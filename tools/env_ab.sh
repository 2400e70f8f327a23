#!/bin/bash
# usage: bash tools/env_ab.sh <tag> <kernel regex> "<ENV A>" "<ENV B>" ...  — same build, variants by environment (tuning keys read from LUMEN_MI_*):
# alone time of the matching kernels (single stream, rocprofv3) + three bench lines per variant; list every variant twice or more to interleave.
tag=$1; shift; kern=$1; shift; mkdir -p gpurun_out/$tag; R=$PWD
for ex in "$@"; do
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 $ex && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact $AB_ARGS > $R/gpurun_out/$tag/prof.log 2>&1)
  f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$kern" "$ex" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]) and not r["Name"].endswith("_inst"): print("[",sys.argv[3],"]", r["Name"], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3))
PY
  for i in 1 2 3; do (export $ex; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact $AB_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('  bench', j['value'], j['ms_per_step'], j['device_ms_per_traceframe'])"); done
done
rm -rf gpurun_out/$tag/prof
