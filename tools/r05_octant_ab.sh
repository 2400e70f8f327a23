#!/bin/bash
# continuation / NEE rays ordered by direction octant inside the block-aggregated append (LM_APPEND_OCTANT) against the plain append:
# interleaved bench runs on one box (prebuilt libraries, tools/ab_lib.sh), the SQ counter pass of both (active lanes per VALU instruction of the traversal kernels),
# and the parity of the variant (full-size C2 + sandbox + Cornell against the oracle).
mkdir -p gpurun_out/r05; out=gpurun_out/r05/append_octant_ab.txt; : > $out
bash tools/ab_lib.sh run $out 4 "--steps 10 --warmup 2" base octant > /dev/null
for v in base octant; do
  LUMEN_MI_LIBRARY=$PWD/lumenrenderer_amd/ab/liblumen_mi_$v.so bash tools/pmc.sh r05/pmc_$v "SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > gpurun_out/r05/pmc_$v.txt 2>&1
  echo "== $v: SQ counters summed over the launches of bench.py --steps 1 (columns: SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES); lanes per VALU instruction = THREAD_CYCLES_VALU / INSTS_VALU / 4... see below" >> $out
  grep -E "^kernel|lm_k_trace_closest |lm_k_trace_shadow|lm_k_shade_wave|lm_k_extract0|lm_k_path_tail" gpurun_out/r05/pmc_$v.txt >> $out
done
python3 - >> $out <<'PY'
import re
for v in ("base", "octant"):
    rows = {}
    hdr = None
    for line in open(f"gpurun_out/r05/pmc_{v}.txt"):
        p = line.split()
        if p and p[0] == "kernel": hdr = p[2:]
        elif p and p[0].startswith("lm_k") and hdr:
            rows[p[0]] = dict(zip(hdr, map(float, p[2:])))
    for k in ("lm_k_trace_closest", "lm_k_trace_shadow", "lm_k_path_tail", "lm_k_path_tail_pair"):
        if k in rows:
            r = rows[k]; ins = [x for n, x in r.items() if n.endswith("INSTS_VALU")][0]; thr = [x for n, x in r.items() if n.endswith("CYCLES_VALU")][0]
            print(f"[{v}] {k}: active lanes per VALU instruction {thr / max(ins, 1.0):.1f}  (VALU wave-instructions {ins:.4g})")
PY
echo "== parity of the octant build (exact mode is order-independent: bit-identical images are the requirement)" >> $out
LUMEN_MI_LIBRARY=$PWD/lumenrenderer_amd/ab/liblumen_mi_octant.so timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "c2_at_full_size or cornell_c1 or sandbox_default or schedules_do_not_change or c3_at_full" 2>&1 | tail -3 >> $out
cat $out
