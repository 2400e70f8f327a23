#!/bin/bash
# usage: bash tools/build_ab.sh <tag> "<EXTRA flags A>" "<EXTRA flags B>" ...  — rebuild the library with each flag set on the GPU box and bench it
tag=$1; shift; mkdir -p gpurun_out/$tag
for ex in "$@"; do
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E "error" 
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$ex" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"])
except Exception as ex: print(sys.argv[1], "failed", ex, open("gpurun_out/$tag/b.err").read()[-800:])
PY
done
