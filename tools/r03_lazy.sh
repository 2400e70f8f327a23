#!/bin/bash
# A/B, interleaved on one box: lazy reuse (tuning key lazy_reuse; the default at even path depths) vs the history passes launched with every frame
mkdir -p gpurun_out/r03_lazy
for rep in 1 2 3; do
  for ex in "LUMEN_MI_LAZY_REUSE=0" "LUMEN_MI_LAZY_REUSE=-1"; do
    (export $ex; python bench.py --steps 10 --warmup 2 --no-cpu-baseline $AB_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$ex]', j['value'], j['ms_per_step'], 'exact', j.get('value_exact'), j.get('ms_per_step_exact'))")
  done
done | tee gpurun_out/r03_lazy/ab.txt
