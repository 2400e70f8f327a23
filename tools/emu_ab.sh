#!/bin/bash
# usage: bash tools/emu_ab.sh <tag> "<EXTRA A>" "<EXTRA B>" ...  — per-rank frame time of the emulated 2- / 4- / 8-rank windows (and the full frame) per build variant
tag=$1; shift; mkdir -p gpurun_out/$tag
for ex in "$@"; do
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
  line="[ $ex ]"
  for e in "" "--emulate-rank 1/2" "--emulate-rank 1/4" "--emulate-rank 1/8"; do
    timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $e > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
    line="$line $(python3 -c "import json;print(json.loads(open('gpurun_out/$tag/b.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo fail)"
  done
  echo "$line   (ms per frame: full, rank of 2, of 4, of 8)"
done
make -C lumenrenderer_amd/csrc clean > /dev/null
