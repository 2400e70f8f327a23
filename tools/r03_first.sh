#!/bin/bash
# round 3, first GPU call: new known-answer tests, VALU issue ceiling, LDS counters of the candidate pick
tag=${1:-r03a}; R=$PWD; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -q -s -k "resample_and_combine or contracted_bsdf or reservoir_cdf or bsdf_matches" 2>&1 | tail -15 > gpurun_out/$tag/pytest_new.log; cat gpurun_out/$tag/pytest_new.log
timeout 300 tools/bin/valu_peak > gpurun_out/$tag/valu_peak.txt 2>&1; cat gpurun_out/$tag/valu_peak.txt
rocprofv3 --list-avail 2>/dev/null | grep -i "lds" | head -60 > gpurun_out/$tag/lds_counters.txt
pass() {  # name, counters...
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$tag/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact > $R/gpurun_out/$tag/$name.log 2>&1)
  f=$(find gpurun_out/$tag/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not k.startswith("lm_k") or k.endswith("_inst"): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    n = len(disp[k]); print(f"{k:36s} x{n:3d} " + "  ".join(f"{c} {v / n:.4g}" for c, v in sorted(agg[k].items())))
PY
  rm -rf gpurun_out/$tag/$name
}
pass lds1 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES > gpurun_out/$tag/pmc_lds.txt 2>&1
pass lds2 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM >> gpurun_out/$tag/pmc_lds.txt 2>&1
pass valu2 SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA >> gpurun_out/$tag/pmc_lds.txt 2>&1
grep -v "^$" gpurun_out/$tag/pmc_lds.txt | grep "pick\|Error\|error\|rror" | head -40
