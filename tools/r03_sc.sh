#!/bin/bash
tag=${1:-r03sc}; mkdir -p gpurun_out/$tag
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
OLD="-DLM_TEMPORAL_SHORTCUT=0 -DLM_COMBINE_SHORTCUT=0 -DLM_GB_PARAMS_FIRST=0"
KREGEX="temporal|combine" bash tools/slab_ab.sh $tag "$OLD" "-DLM_COMBINE_SHORTCUT=0 -DLM_GB_PARAMS_FIRST=0" "-DLM_GB_PARAMS_FIRST=0" "-DLM_R3=1" "$OLD" "-DLM_R3=1" "$OLD" "-DLM_R3=1" 2>&1 | tee gpurun_out/$tag/ab.txt
