#!/bin/bash
# usage: bash tools/emu.sh <tag>  — per-rank time of the tiled multi-GPU path, emulated on one GPU (rank 1 = an interior-column tile)
tag=${1:-emu}; mkdir -p gpurun_out/$tag
for e in "1/2" "1/4" "1/8" "1/2 --reuse lazy" "1/4 --reuse lazy" "1/8 --reuse lazy"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --emulate-rank $e > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$e" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "tile-Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["emulated_rank"]["window"], j["device_ms_per_traceframe"])
except Exception as ex: print("failed", ex, open("gpurun_out/$tag/b.err").read()[-1500:])
PY
done
