#!/bin/bash
# round 4: device tree build (tuning key gpu_build) — parity tests, then build time and frame rate against the host SAH builder on C2 (262 k triangles) and C5 (10 M)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_tree_build or (schedules_do_not and gpu_build)" 2>&1 | tail -4
for wl in c2 c5; do
  for g in 0 1; do
    echo "== workload $wl gpu_build $g"
    LUMEN_MI_GPU_BUILD=$g LUMEN_MI_BUILD_TIMING=1 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse 2> gpurun_out/gb.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  Mrays/s', j['value'], 'ms/step', j['ms_per_step'], 'nodes4/ray', j['config']['nodes4_per_ray'], 'tris/ray', j['config']['tris_per_ray'])"
    grep '\[bvh\]' gpurun_out/gb.err | tail -12
  done
done
