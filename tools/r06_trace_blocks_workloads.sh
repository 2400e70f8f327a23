#!/bin/bash
# usage (GPU box): bash tools/r06_trace_blocks_workloads.sh — the per-frame choice of the persistent traversal grids (csrc/frame.cpp: half the grid for the visibility passes of
# short light lists and for the primary launch of eager frames on large windows / cacheable trees) against the full grid everywhere, every workload, variants interleaved
# (profiles/r06_trace_blocks_ab.txt)
mkdir -p gpurun_out/r6x
full="LUMEN_MI_TRACE_BLOCKS_MAIN=8 LUMEN_MI_TRACE_BLOCKS_VIS=8"
for w in c2 c2t c3 c4 c5; do
  bash tools/ab_knob.sh gpurun_out/r6x/tb3_$w.txt 3 "--workload $w --steps 6 --warmup 2" "$full" "-" > gpurun_out/r6x/tb3_$w.log 2>&1
  echo "== $w"; grep -v 'device ms\|^#' gpurun_out/r6x/tb3_$w.txt
done
bash tools/ab_knob.sh gpurun_out/r6x/tb3_lowpoly.txt 3 "--workload lowpoly --steps 48 --warmup 8" "$full" "-" > gpurun_out/r6x/tb3_lowpoly.log 2>&1; echo "== lowpoly"; grep -v 'device ms\|^#' gpurun_out/r6x/tb3_lowpoly.txt
bash tools/ab_knob.sh gpurun_out/r6x/tb3_sandbox.txt 3 "--workload sandbox --steps 32 --warmup 8" "$full" "-" > gpurun_out/r6x/tb3_sandbox.log 2>&1; echo "== sandbox"; grep -v 'device ms\|^#' gpurun_out/r6x/tb3_sandbox.txt
