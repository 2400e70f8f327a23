#!/bin/bash
# lazy reuse: the wave chain is the critical path now — stream priorities, residency caps (interleaved on one box).  (The recorded run also had two
# temporary switches, the shadow / tail stream at default priority and 4 / 6 instead of 8 tail blocks per CU: both neutral, removed again.)
mkdir -p gpurun_out/r03_lazy
run() { (export $1; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse --reuse lazy 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$1]', j['value'], j['ms_per_step'])"); }
for rep in 1 2; do
  for ex in "LUMEN_MI_NOP=1" "LUMEN_MI_AUX_PRIORITY=0" "LUMEN_MI_AUX3_PRIORITY=1" "LUMEN_MI_TAIL_LANES=32" "LUMEN_MI_CAP_TEMPORAL=4" "LUMEN_MI_CAP_EXTRACT=4" "LUMEN_MI_CAP_EXTRACT=6" "LUMEN_MI_PACKET_PRIMARY=0" "LUMEN_MI_REFILL=0"; do run "$ex"; done
done | tee gpurun_out/r03_lazy/knobs2.txt
