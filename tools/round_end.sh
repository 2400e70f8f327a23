#!/bin/bash
# usage (GPU box, repo root): bash tools/round_end.sh <tag>  — everything the round's record needs in one call: GPU parity suite, bench lines
# (default fast mode with the exact mode beside it; every workload), rocprofv3 kernel stats in both modes, the three PMC passes, per-rank
# emulation.  Outputs under gpurun_out/<tag>/; the builder copies the summaries to profiles/.
tag=${1:-round_end}; R=$PWD; mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
python bench.py > gpurun_out/$tag/bench_c2.json 2> gpurun_out/$tag/bench_c2.err; tail -1 gpurun_out/$tag/bench_c2.json | cut -c1-300
timeout 900 python bench.py --workload sandbox --steps 32 --warmup 8 > gpurun_out/$tag/bench_sandbox.json 2> gpurun_out/$tag/bench_sandbox.err; tail -1 gpurun_out/$tag/bench_sandbox.json | cut -c1-300
timeout 900 python bench.py --workload lowpoly --steps 64 --warmup 8 > gpurun_out/$tag/bench_lowpoly.json 2> gpurun_out/$tag/bench_lowpoly.err; tail -1 gpurun_out/$tag/bench_lowpoly.json | cut -c1-300
for w in c2t c3 c4 c5 c1; do
  timeout 900 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err
  python3 - "$w" "gpurun_out/$tag/bench_$w.json" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fast", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame | exact", j["config"]["other_mode"]["value"], j["config"]["other_mode"]["ms_per_step"],
          "| lazy reuse: fast", j["value_lazy_reuse"], j["ms_per_step_lazy_reuse"], "exact", j["value_exact_lazy_reuse"])
except Exception as ex: print(sys.argv[1], "failed", ex)
PY
done
for mode in fast exact lazy_fast lazy_exact; do
  margs="--mode ${mode#lazy_} --reuse eager"; case $mode in lazy_*) margs="--mode ${mode#lazy_} --reuse lazy";; esac
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof_$mode -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse $margs > $R/gpurun_out/$tag/prof_$mode.log 2>&1)
  f=$(find gpurun_out/$tag/prof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$tag/kernel_stats_$mode.csv
  t=$(find gpurun_out/$tag/prof_$mode -name "*kernel_trace.csv" | head -1)
  echo "--- $mode: per-kernel averages under overlap, then the steady-state timeline"
  python3 - gpurun_out/$tag/kernel_stats_$mode.csv <<'PY'
import csv,sys
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:16]: print(f'{r["Name"][:34]:34s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
  python3 tools/timeline.py "$t" > gpurun_out/$tag/timeline_$mode.txt; rm -rf gpurun_out/$tag/prof_$mode
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/alone_$mode -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse $margs > $R/gpurun_out/$tag/alone_$mode.log 2>&1)
  f=$(find gpurun_out/$tag/alone_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$tag/kernel_stats_alone_$mode.csv; rm -rf gpurun_out/$tag/alone_$mode
done
bash tools/pmc_json.sh $tag/pmc > gpurun_out/$tag/pmc_summary.txt 2>&1; tail -30 gpurun_out/$tag/pmc_summary.txt
bash tools/emu.sh $tag/emu 2>&1 | tail -7
