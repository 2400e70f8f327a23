#!/bin/bash
tag=${1:-r03v}; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q -k "schedules or early_wave or fast_resampling or c2_at_full or fuzz" 2>&1 | tail -5 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/env_ab.sh $tag "restir_trace" "LUMEN_MI_PACKET_VISIBILITY=0" "LUMEN_MI_PACKET_VISIBILITY=1" "LUMEN_MI_PACKET_VISIBILITY=0" "LUMEN_MI_PACKET_VISIBILITY=1" 2>&1 | tee gpurun_out/$tag/ab.txt
AB_ARGS="--workload c3" bash tools/env_ab.sh $tag "restir_trace" "LUMEN_MI_PACKET_VISIBILITY=0" "LUMEN_MI_PACKET_VISIBILITY=1" 2>&1 | tee gpurun_out/$tag/ab_c3.txt
