#!/bin/bash
# usage: bash tools/ab2.sh <tag> "<bench args>" "ENV1=a" "ENV2=b" ...
tag=$1; shift; args=$1; shift; mkdir -p gpurun_out/$tag
for cfg in "" "$@"; do
  env $cfg timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$cfg" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"], j["config"].get("nodes4_per_ray"))
except Exception as ex: print(sys.argv[1], "failed", ex, open("gpurun_out/$tag/b.err").read()[-800:])
PY
done
