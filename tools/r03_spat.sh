#!/bin/bash
tag=${1:-r03s}; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q -k "probes_in_lds or fast_resampling" 2>&1 | tail -5 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/env_ab.sh $tag "spatial" "LUMEN_MI_SPATIAL_LDS=0" "LUMEN_MI_SPATIAL_LDS=1" "LUMEN_MI_SPATIAL_LDS=0" "LUMEN_MI_SPATIAL_LDS=1" 2>&1 | tee gpurun_out/$tag/ab.txt
