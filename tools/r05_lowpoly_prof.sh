#!/bin/bash
# the reference's default model as a workload: bench line + per-kernel device time under overlap and alone (rocprofv3 --kernel-trace --stats), fast and exact
R=$PWD; mkdir -p gpurun_out/r05
timeout 600 python bench.py --workload lowpoly --steps 64 --warmup 8 > gpurun_out/r05/bench_lowpoly.json 2> gpurun_out/r05/bench_lowpoly.err; tail -c 600 gpurun_out/r05/bench_lowpoly.json
timeout 600 python bench.py --workload sandbox --steps 64 --warmup 8 > gpurun_out/r05/bench_sandbox.json 2> gpurun_out/r05/bench_sandbox.err
for wl in lowpoly sandbox; do for mode in fast; do
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof_${wl}_$mode -- python3 $R/bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse --mode $mode > $R/gpurun_out/r05/prof_${wl}_$mode.log 2>&1)
  f=$(find gpurun_out/r05/prof_${wl}_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/${wl}_kernel_stats_$mode.csv; rm -rf gpurun_out/r05/prof_${wl}_$mode
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/alone_${wl}_$mode -- python3 $R/bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse --mode $mode > $R/gpurun_out/r05/alone_${wl}_$mode.log 2>&1)
  f=$(find gpurun_out/r05/alone_${wl}_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/${wl}_kernel_stats_alone_$mode.csv; rm -rf gpurun_out/r05/alone_${wl}_$mode
  echo "--- $wl $mode: per-kernel averages alone (single stream)"
  python3 - gpurun_out/r05/${wl}_kernel_stats_alone_$mode.csv <<'PY'
import csv,sys
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:14]: print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
done; done
