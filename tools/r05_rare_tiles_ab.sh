#!/bin/bash
# the tile map of surfaces that need the exact launch of the fast ReSTIR passes (LM_RARE_TILES) against every block looking at its pixels: LowpolyRoom (0.08 % glass) and the atrium with one glass object
mkdir -p gpurun_out/r05; out=gpurun_out/r05/rare_tiles_ab.txt; : > $out
bash tools/ab_lib.sh run $out 4 "--workload lowpoly --steps 64 --warmup 8 --no-other-reuse" nomap base > /dev/null
for round in 1 2; do for v in nomap base; do LUMEN_MI_LIBRARY=$PWD/lumenrenderer_amd/ab/liblumen_mi_$v.so python tools/r05_rare_tiles.py 2>/dev/null | tail -1 >> $out; done; done
echo "== parity of the build with the map (fast mode against the oracle on scenes with glass / clear coat; LowpolyRoom; tiles)" >> $out
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_lowpoly.py tests/test_gpu_parity.py -k "lowpoly or fast_resampling or fast_mode_with or history_passes_run or stitched_tiles or seam_history or fast_policy" 2>&1 | tail -3 >> $out
cat $out
