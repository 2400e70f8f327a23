#!/bin/bash
# A/B of the LDS-staged top of the tree (LM_TOP_NODES) and the LDS share of the traversal stack (LM_STACK_LDS): whole-frame bench in both
# modes + alone time of the traversal kernels, per build variant.  usage (GPU box): bash tools/top_ab.sh <tag>
tag=${1:-top_ab}; mkdir -p gpurun_out/$tag; R=$PWD
for ex in "-DLM_TOP_NODES=0" "-DLM_TOP_NODES=5" "-DLM_TOP_NODES=21" "-DLM_TOP_NODES=85 -DLM_STACK_LDS=12" "-DLM_TOP_NODES=85" "-DLM_TOP_NODES=0 -DLM_STACK_LDS=12"; do
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
  for rep in 1 2; do
    timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
    python3 - "$ex" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] fast", j["value"], "exact", j["config"]["other_mode"]["value"], "closest launch_ms", j["roofline"]["launch_ms"])
except Exception as ex: print(sys.argv[1], "failed", ex, open("gpurun_out/$tag/b.err").read()[-800:])
PY
  done
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > $R/gpurun_out/$tag/prof.log 2>&1)
  f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$ex" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "trace_" in r["Name"] and not r["Name"].endswith("_inst"): print("    alone [",sys.argv[2],"]", r["Name"], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3))
PY
done
rm -rf gpurun_out/$tag/prof
make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 2>&1 | grep -E " error"
