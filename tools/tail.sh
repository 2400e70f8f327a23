#!/bin/bash
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "LUMEN_MI_TAIL_BELOW=0" "LUMEN_MI_TAIL_BELOW=32768" "LUMEN_MI_TAIL_BELOW=65536" "LUMEN_MI_TAIL_BELOW=131072" "LUMEN_MI_TAIL_BELOW=65536 LUMEN_MI_TAIL_LANES=32"; do
 for e in "" "--emulate-rank 1/8" "--emulate-rank 1/4"; do
  env $cfg timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $e > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
  python3 - "$cfg $e" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "| Mrays/s", j["value"], "ms/frame", j["ms_per_step"], j["device_ms_per_traceframe"])
except Exception as ex: print(sys.argv[1], "failed", ex, open("gpurun_out/$tag/b.err").read()[-800:])
PY
 done
done
