#!/bin/bash
# usage: bash tools/cfgs.sh <tag>  — one bench line per non-default workload (c3, c4, c5) for the record
tag=${1:-cfg}
mkdir -p gpurun_out/$tag
for w in c3 c4 c5; do
  t0=$(date +%s)
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err
  tail -1 gpurun_out/$tag/bench_$w.json | cut -c1-900

  echo "$w wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/$tag/bench_$w.err
done
