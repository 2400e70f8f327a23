#!/bin/bash
# knob sweep on the end-of-round build (fast mode, C2): each setting 3 bench lines, the default three times (first, middle, last)
tag=${1:-r03k}; mkdir -p gpurun_out/$tag
run() { for i in 1 2 3; do (export $1; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'])"); done; }
{ run A=1; run LUMEN_MI_TAIL_BELOW=100000; run LUMEN_MI_TAIL_BELOW=250000; run LUMEN_MI_TAIL_BELOW=32768; run LUMEN_MI_PICK_AHEAD=0; run A=2
  run LUMEN_MI_SHADOW_ON_WAVE=1; run LUMEN_MI_WAVE_STREAMS=2; run LUMEN_MI_TAIL_PAIR=1; run LUMEN_MI_REFILL=0; run LUMEN_MI_REFILL_VIS=0; run LUMEN_MI_AUX3_PRIORITY=1; run LUMEN_MI_AUX_PRIORITY=0; run A=3; } 2>&1 | tee gpurun_out/$tag/knobs.txt
