#!/bin/bash
# build variants (EXTRA flags) side by side: alone time of the kernels matching $KREGEX (default: traversal) + three bench lines per build
tag=${1:-slab_ab}; shift; mkdir -p gpurun_out/$tag
for ex in "$@"; do
  KAB_ARGS="--no-exact" bash tools/kernel_ab.sh $tag "${KREGEX:-trace|tail|query}" "$ex"
  for i in 1 2 3; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('  bench', j['value'], j['ms_per_step'], j['device_ms_per_traceframe'])"; done
done
make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 2>&1 | grep -E " error"
