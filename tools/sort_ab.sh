#!/bin/bash
# A/B of the ray reorder between waves (tuning key sort_rays = deepest wave whose queue is sorted): whole-frame bench + alone times of the
# closest-hit launches and of the sort kernels.  usage (GPU box): bash tools/sort_ab.sh <tag>
tag=${1:-sort_ab}; mkdir -p gpurun_out/$tag; R=$PWD
for v in 0 1 2 5; do
  for rep in 1 2; do
    LUMEN_MI_SORT_RAYS=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err
    python3 - "sort_rays=$v" <<PY
import json,sys
try:
    j=json.loads(open("gpurun_out/$tag/b.json").read().strip().splitlines()[-1])
    print("[",sys.argv[1],"] fast", j["value"], "exact", j["config"]["other_mode"]["value"], "closest launch_ms", j["roofline"]["launch_ms"], j["device_ms_per_traceframe"])
except Exception as ex: print(sys.argv[1], "failed", ex, open("gpurun_out/$tag/b.err").read()[-800:])
PY
  done
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 LUMEN_MI_TAIL_BELOW=0 LUMEN_MI_SORT_RAYS=$v && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > $R/gpurun_out/$tag/prof.log 2>&1)
  t=$(find gpurun_out/$tag/prof -name "*kernel_trace.csv" | head -1)
  python3 - "$t" "sort_rays=$v" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("lm_k") and not r["Kernel_Name"].endswith("_inst")]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# per TraceFrame: the closest-hit launches in order (wave 0, 1, ...), averaged over the last 8 TraceFrames; sort kernels summed
frames=[]; cur=None
for r in rows:
    if r["Kernel_Name"]=="lm_k_primary": cur={"closest":[], "sort":0.0}; frames.append(cur)
    if cur is None: continue
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if r["Kernel_Name"]=="lm_k_trace_closest": cur["closest"].append(d)
    if r["Kernel_Name"].startswith("lm_k_sort"): cur["sort"]+=d
fr=[f for f in frames[-8:] if len(f["closest"])==len(frames[-1]["closest"])]
n=len(fr[0]["closest"])
print("    alone [",sys.argv[2],"] closest-hit us by wave:", " ".join("%.0f"%(sum(f["closest"][k] for f in fr)/len(fr)) for k in range(n)), "| sum %.0f"%(sum(sum(f["closest"]) for f in fr)/len(fr)), "| sort kernels %.0f"%(sum(f["sort"] for f in fr)/len(fr)))
PY
done
rm -rf gpurun_out/$tag/prof
