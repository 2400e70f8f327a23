#!/bin/bash
tag=${1:-emu}; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for e in "" "--emulate-rank 1/2" "--emulate-rank 1/4" "--emulate-rank 1/8"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $e > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$e" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"])
except Exception as ex: print("failed", ex, open("gpurun_out/$tag/b.err").read()[-1500:])
PY
done
