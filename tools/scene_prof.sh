#!/bin/bash
# usage (GPU box): bash tools/scene_prof.sh <scene.npz> [W H depth] — per-kernel ALONE averages (single stream) of tools/scene_ms.py on a scene file, under rocprofv3 --kernel-trace --stats
R=$PWD; npz=$(readlink -f $1); shift; mkdir -p gpurun_out/scene_prof; rm -rf gpurun_out/scene_prof/p
(cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 LUMEN_MI_SINGLE_STREAM=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scene_prof/p -- python3 $R/tools/scene_ms.py $npz "$@" > $R/gpurun_out/scene_prof/log.txt 2>&1)
tail -2 gpurun_out/scene_prof/log.txt | cut -c1-300
f=$(find gpurun_out/scene_prof/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:24]:
    print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} pct {r["Percentage"]}')
PY
rm -rf gpurun_out/scene_prof/p
