#!/bin/bash
tag=${1:-r03w}; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/slab_ab.sh $tag "-DLM_WIDTH=4" "-DLM_WIDTH=8" "-DLM_WIDTH=8 -DLM_TRACE_WAVES=6" "-DLM_WIDTH=4" "-DLM_WIDTH=8" 2>&1 | tee gpurun_out/$tag/ab.txt
