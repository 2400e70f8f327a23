#!/bin/bash
# usage (GPU box, repo root): bash tools/robust.sh <tag>  — robustness record of a build: schedule-fuzz campaign (exact, fast, two wave streams) + soak
tag=${1:-robust}; mkdir -p gpurun_out/$tag
timeout 3000 python tools/fuzz.py ${FUZZ_SEEDS:-40} 2>&1 | tail -4 | tee gpurun_out/$tag/fuzz_exact.txt
LUMEN_MI_FAST_RESAMPLE=1 timeout 2000 python tools/fuzz.py ${FUZZ_SEEDS_FAST:-20} 2>&1 | tail -4 | tee gpurun_out/$tag/fuzz_fast.txt
LUMEN_MI_WAVE_STREAMS=2 timeout 2000 python tools/fuzz.py ${FUZZ_SEEDS_FAST:-20} 2>&1 | tail -4 | tee gpurun_out/$tag/fuzz_ws2.txt
timeout 1200 python tools/soak.py 24000 2>&1 | tail -6 | tee gpurun_out/$tag/soak.txt
