#!/bin/bash
tag=${1:-r03p}; mkdir -p gpurun_out/$tag
timeout 300 tools/bin/valu_peak > gpurun_out/$tag/valu_peak.txt 2>&1; grep -A200 "fma pairs" gpurun_out/$tag/valu_peak.txt | grep " 8 "
bash tools/slab_ab.sh $tag "-DLM_PERM_VGPR=0" "-DLM_PERM_VGPR=1" "-DLM_PERM_VGPR=0" "-DLM_PERM_VGPR=1" 2>&1 | tee gpurun_out/$tag/ab.txt
