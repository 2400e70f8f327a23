#!/bin/bash
# usage (GPU box): bash tools/r06_overlap_prof.sh OUT LIB_A LIB_B ... — per-kernel averages UNDER OVERLAP (the frame's four streams live; rocprofv3 --kernel-trace --stats) of C2,
# fast mode, eager history passes, for prebuilt library variants: which kernels stretch when their neighbours change (round 6: profiles/r06_slp_overlap.txt)
out=$1; shift; : > $out; R=$PWD
for v in "$@"; do
  export LUMEN_MI_LIBRARY=$R/lumenrenderer_amd/ab/liblumen_mi_$v.so
  rm -rf gpurun_out/ovl_prof
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ovl_prof -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact --no-other-reuse > $R/gpurun_out/ovl_prof.log 2>&1)
  f=$(find gpurun_out/ovl_prof -name "*kernel_stats.csv" | head -1)
  echo "#### $v (overlapped, fast eager); bench line: $(tail -1 gpurun_out/ovl_prof.log | cut -c1-120)" >> $out
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:16]:
    print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f}')
PY
  rm -rf gpurun_out/ovl_prof
done
unset LUMEN_MI_LIBRARY
