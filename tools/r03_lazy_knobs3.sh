#!/bin/bash
# exact mode: the path-tail threshold re-checked with lazy reuse on (and with the history passes in every frame), interleaved on one box
mkdir -p gpurun_out/r03_lazy
run() { (export $1; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --mode exact 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$1] exact: eager', j['value_eager_reuse'], 'lazy', j['value_lazy_reuse'], j['ms_per_step_lazy_reuse'])"); }
for rep in 1 2; do for ex in "LUMEN_MI_NOP=1" "LUMEN_MI_TAIL_BELOW=65536" "LUMEN_MI_TAIL_BELOW=100000" "LUMEN_MI_TAIL_BELOW=200000"; do run "$ex"; done; done | tee gpurun_out/r03_lazy/knobs3.txt
