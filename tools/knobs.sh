#!/bin/bash
# usage: bash tools/knobs.sh <tag> "<bench args>" "ENV=.. ENV=.." ...  — bench line per environment setting (knobs of renderer.cpp), two rounds
tag=$1; shift; args=$1; shift; mkdir -p gpurun_out/$tag
for round in 1 2; do
for e in "$@"; do
  env $e timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$e" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"])
except Exception as ex: print("failed", ex, open("gpurun_out/$tag/b.err").read()[-1500:])
PY
done
done
