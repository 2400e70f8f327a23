#!/bin/bash
# usage: bash tools/pmc_ab.sh <tag> "<EXTRA flags>" ...  — VALU counters of the ReSTIR kernels per build variant (serialised kernels)
tag=$1; shift; mkdir -p gpurun_out/$tag; R=$PWD
for ex in "$@"; do
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$tag/pmc.log 2>&1)
  f=$(find gpurun_out/$tag/prof -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$ex" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if k.endswith("_inst") or not k.startswith("lm_k"): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
print("[", sys.argv[2], "]")
for k in ("lm_k_pick_primary","lm_k_restir_combine","lm_k_restir_temporal","lm_k_restir_spatial"):
    a=agg[k]; n=max(1,len(calls[k]))
    if not a: continue
    print("   %-26s calls %2d  insts/call %.3e  valu_busy %.2f  lanes %.2f  gpu_us/call %.0f" % (k, n, a["SQ_INSTS_VALU"]/n, a["SQ_INSTS_VALU"]*4/(a["GRBM_GUI_ACTIVE"]*128), a["SQ_THREAD_CYCLES_VALU"]/(a["SQ_INSTS_VALU"]*64), a["GRBM_GUI_ACTIVE"]/8/n/2400))
PY
done
rm -rf gpurun_out/$tag/prof
