#!/bin/bash
# path-tail shape under the fast ReSTIR mode: paths per wavefront (tail_lanes) x threshold, whole frame and one rank of 8
run() { echo -n "$* : "; env "${@:2}" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact $1 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['device_ms_per_traceframe'])"; }
for a in "" "--emulate-rank 1/8"; do
  for l in 16 32 64; do
    run "$a" LUMEN_MI_TAIL_LANES=$l
  done
  run "$a" LUMEN_MI_TAIL_LANES=32 LUMEN_MI_TAIL_BELOW=100000
  run "$a" LUMEN_MI_TAIL_LANES=64 LUMEN_MI_TAIL_BELOW=250000
  run "$a" LUMEN_MI_TAIL_LANES=8
done
