#!/bin/bash
# A/B on one box: the history passes launched with every frame (lazy_reuse 0: the bench headline) vs lazy reuse (the default at even path depths).
# bench.py times both settings in one run, fast mode then exact mode, interleaved: eager fast, eager exact, lazy fast, lazy exact.
mkdir -p gpurun_out/r03_lazy
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $AB_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[eager]', j['value_eager_reuse'], 'exact', j['value_exact_eager_reuse'], '[lazy]', j['value_lazy_reuse'], j['ms_per_step_lazy_reuse'], 'exact', j['value_exact_lazy_reuse'])"
done | tee gpurun_out/r03_lazy/ab.txt
