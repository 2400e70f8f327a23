#!/bin/bash
# usage: bash tools/ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   — one bench line per environment setting (A/B runs on one box)
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for cfg in "" "$@"; do
  env $cfg timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_$i.json 2> gpurun_out/$tag/bench_$i.err
  python3 - <<PY
import json
try:
    j=json.loads(open("gpurun_out/$tag/bench_$i.json").read().strip().splitlines()[-1])
    print("[$cfg]", "Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"])
except Exception as e: print("[$cfg] failed", e, open("gpurun_out/$tag/bench_$i.err").read()[-1500:])
PY
  i=$((i+1))
done
