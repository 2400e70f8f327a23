#!/bin/bash
# usage (GPU box): bash tools/r06_slp_prof.sh OUT LIB_A LIB_B ... — per-kernel ALONE averages (single stream, rocprofv3 --kernel-trace --stats) of C2 in the fast and the exact mode for
# prebuilt library variants (tools/ab_lib.sh build): which kernels gain and which lose under a compile-time switch (round 6: -fno-slp-vectorize, profiles/r06_slp_per_kernel.txt)
out=$1; shift; : > $out
for v in "$@"; do
  for mode in fast exact; do
    export LUMEN_MI_LIBRARY=$PWD/lumenrenderer_amd/ab/liblumen_mi_$v.so
    echo "#### $v, mode $mode" >> $out
    bash tools/wl_prof.sh c2 --mode $mode >> $out 2>&1
  done
done
unset LUMEN_MI_LIBRARY
