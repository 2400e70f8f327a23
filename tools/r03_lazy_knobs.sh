#!/bin/bash
# knob sweep with lazy reuse on (the frame is 27 % shorter: does the balance between the streams still hold?), interleaved on one box
mkdir -p gpurun_out/r03_lazy
run() { (export $1; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse --reuse lazy 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$1]', j['value'], j['ms_per_step'])"); }
for rep in 1 2; do
  for ex in "LUMEN_MI_NOP=1" "LUMEN_MI_TAIL_BELOW=16384" "LUMEN_MI_TAIL_BELOW=65536" "LUMEN_MI_TAIL_BELOW=200000" "LUMEN_MI_TAIL_BELOW=400000" "LUMEN_MI_PICK_AHEAD=1" "LUMEN_MI_SHADOW_ON_WAVE=1" "LUMEN_MI_WAVE_STREAMS=2" "LUMEN_MI_TRACE_BLOCKS_MAIN=6" "LUMEN_MI_TRACE_BLOCKS_AUX=6" "LUMEN_MI_TRACE_BLOCKS_AUX=4" "LUMEN_MI_TAIL_PAIR=1" "LUMEN_MI_FAST_SHADE=1" "LUMEN_MI_REFILL_VIS=0"; do run "$ex"; done
done | tee gpurun_out/r03_lazy/knobs.txt
