#!/bin/bash
# usage: bash tools/pmc.sh <tag> "<counter list>"   (GPU box; separate pass per counter group)
tag=${1:-pmc}; ctrs=${2:-"SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"}
R=$PWD; mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact > $R/gpurun_out/$tag/pmc.log 2>&1
cd $R
f=$(find gpurun_out/$tag/prof -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
seen=set()
for r in rows:
    k=r["Kernel_Name"]
    if k.endswith("_inst") or not k.startswith("lm_k"): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(r["Dispatch_Id"],k)
    if key not in seen: seen.add(key); calls[k]+=1
names=sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(32), "calls", " ".join(n[-18:].rjust(18) for n in names))
for k in sorted(agg, key=lambda k:-agg[k].get("SQ_WAVE_CYCLES", agg[k].get(names[0],0))):
    print(k[:32].ljust(32), str(calls[k]).rjust(5), " ".join(f"{agg[k].get(n,0):18.4g}" for n in names))
PY
rm -rf gpurun_out/$tag/prof
