#!/bin/bash
# A/B, interleaved on one box: primary rays generated inside the packet kernel of the primary wave (fuse_primary 1, default) vs their own launch first
mkdir -p gpurun_out/r03_fuse
for rep in 1 2 3; do
  for ex in "LUMEN_MI_FUSE_PRIMARY=0" "LUMEN_MI_FUSE_PRIMARY=1"; do
    (export $ex; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact $AB_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$ex]', j['value'], j['ms_per_step'], 'lazy', j['value_lazy_reuse'], j['ms_per_step_lazy_reuse'])")
  done
done | tee gpurun_out/r03_fuse/ab.txt
