#!/bin/bash
tag=${1:-r03f}; mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -m gpu -x -q -s -k "fast_shade" 2>&1 | grep -E "fast_shade .*rel-L2|passed|failed" > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/env_ab.sh $tag "shade_wave|path_tail" "LUMEN_MI_FAST_SHADE=0" "LUMEN_MI_FAST_SHADE=1" "LUMEN_MI_FAST_SHADE=0" "LUMEN_MI_FAST_SHADE=1" "LUMEN_MI_FAST_SHADE=0" "LUMEN_MI_FAST_SHADE=1" 2>&1 | tee gpurun_out/$tag/ab.txt
