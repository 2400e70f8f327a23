#!/bin/bash
tag=${1:-r03t}; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q -k "schedules or early_wave or fuzz or pipelined" 2>&1 | tail -5 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/env_ab.sh $tag "path_tail" "LUMEN_MI_TAIL_PAIR=0" "LUMEN_MI_TAIL_PAIR=1" "LUMEN_MI_TAIL_PAIR=0" "LUMEN_MI_TAIL_PAIR=1" 2>&1 | tee gpurun_out/$tag/ab.txt
for e in "1/8" "1/4" "1/2"; do
  AB_ARGS="--emulate-rank $e" bash tools/env_ab.sh $tag "path_tail" "LUMEN_MI_TAIL_PAIR=0" "LUMEN_MI_TAIL_PAIR=1" "LUMEN_MI_TAIL_PAIR=0" "LUMEN_MI_TAIL_PAIR=1" 2>&1 | sed "s|^|rank $e |" | tee -a gpurun_out/$tag/ab_ranks.txt
done
