#!/bin/bash
tag=${1:-r03bv}; mkdir -p gpurun_out/$tag
run() { (export $1 $2; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact $BARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1 $2', j['value'], j['ms_per_step'], 'nodes/ray', c['nodes4_per_ray'], 'tris/ray', c['tris_per_ray'])"); }
{ for rep in 1 2 3; do run LUMEN_MI_BVH_SWEEP=0; run LUMEN_MI_BVH_SWEEP=16; run LUMEN_MI_BVH_SWEEP=64; run LUMEN_MI_BVH_SWEEP=512; run LUMEN_MI_BVH_SWEEP=4096; done; } 2>&1 | tee gpurun_out/$tag/ab.txt
