#!/bin/bash
# rocprofv3 kernel trace of the bench with lazy reuse: per-kernel averages under overlap + the steady-state timeline (tools/timeline.py)
R=$PWD; tag=r03_lazy; mkdir -p gpurun_out/$tag
for mode in fast exact; do
  rm -rf gpurun_out/$tag/prof_$mode
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof_$mode -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse --reuse lazy --mode $mode > $R/gpurun_out/$tag/prof_$mode.log 2>&1)
  f=$(find gpurun_out/$tag/prof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$tag/kernel_stats_lazy_$mode.csv
  t=$(find gpurun_out/$tag/prof_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py "$t" > gpurun_out/$tag/timeline_lazy_$mode.txt; rm -rf gpurun_out/$tag/prof_$mode
done
head -70 gpurun_out/$tag/timeline_lazy_fast.txt
