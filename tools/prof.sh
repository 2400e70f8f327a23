#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag>  — parity tests, bench line, rocprofv3 kernel stats
tag=${1:-x}
mkdir -p gpurun_out/$tag
# SKIP_TESTS=1: profile only (e.g. with LUMEN_MI_FAST_RESAMPLE=1 exported, which the bit-exact suite must not run under)
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/$tag/pytest.log
if ! grep -q " passed" gpurun_out/$tag/pytest.log || grep -q "failed\|error\|Aborted\|dumped" gpurun_out/$tag/pytest.log; then echo "PARITY TESTS FAILED"; cat gpurun_out/$tag/pytest.log; exit 1; fi
else echo "tests skipped" > gpurun_out/$tag/pytest.log; fi
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$tag/kernel_stats.csv 2>/dev/null
cat gpurun_out/$tag/pytest.log
python3 - <<PY
import json,csv
try:
    j=json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
    print("Mrays/s",j["value"],"ms/frame",j["ms_per_step"],"dev",j["device_ms_per_traceframe"],"roof",j["roofline"]["frac"], j["config"].get("nodes4_per_ray"), "other", j["config"].get("other_mode"))
except Exception as e: print("bench parse failed",e, open("gpurun_out/$tag/bench.err").read()[-2000:])
rows=[r for r in csv.DictReader(open("gpurun_out/$tag/kernel_stats.csv")) if not r["Name"].endswith("_inst")]
for r in rows[:14]: print(f'{r["Name"][:34]:34s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
import glob, subprocess
tf=glob.glob("gpurun_out/$tag/prof/*/*kernel_trace.csv")
if tf:
    print("--- steady-state timeline, two TraceFrames (start_us dur_us | one column per stream)")
    subprocess.run(["python3","tools/timeline.py",tf[0]])
PY
rm -rf gpurun_out/$tag/prof
