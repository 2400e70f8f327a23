#!/bin/bash
tag=${1:-r03e}; R=$PWD; mkdir -p gpurun_out/$tag
bash tools/emu.sh $tag/emu 2>&1 | tail -4
for e in "1/8" "1/4"; do
  n=${e/\//of}
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --emulate-rank $e > $R/gpurun_out/$tag/prof_$n.log 2>&1)
  t=$(find gpurun_out/$tag/prof_$n -name "*kernel_trace.csv" | head -1)
  echo "--- timeline rank $e"; python3 tools/timeline.py "$t" | tee gpurun_out/$tag/timeline_$n.txt | head -70
  rm -rf gpurun_out/$tag/prof_$n
done
