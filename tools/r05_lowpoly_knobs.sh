#!/bin/bash
# scheduling knobs on the reference's default model (the defaults were tuned on the stand-in atrium): same build, variants by environment, interleaved rounds
mkdir -p gpurun_out/r05; out=gpurun_out/r05/lowpoly_knobs.txt; : > $out
variants=("A=1" "LUMEN_MI_TAIL_BELOW=0" "LUMEN_MI_TAIL_BELOW=32768" "LUMEN_MI_TAIL_BELOW=131072" "LUMEN_MI_TAIL_BELOW=320000" "LUMEN_MI_PACKET_VISIBILITY=1" "LUMEN_MI_REFILL_VIS=0" "LUMEN_MI_TAIL_PAIR=0" "LUMEN_MI_PICK_AHEAD=0" "LUMEN_MI_WAVE_STREAMS=2" "LUMEN_MI_TRACE_BLOCKS_AUX=4" "LUMEN_MI_TRACE_BLOCKS_MAIN=4" "LUMEN_MI_PACKET_PRIMARY=0" "LUMEN_MI_FUSE_PRIMARY=1" "LUMEN_MI_SHADOW_ON_WAVE=1" "LUMEN_MI_AUX3_PRIORITY=1")
for round in 1 2 3; do for v in "${variants[@]}"; do
  line=$(export $v; timeout 120 python bench.py --workload ${WL:-lowpoly} --steps 64 --warmup 8 --no-cpu-baseline --no-exact 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "$v $line" >> $out
done; done
python3 - $out <<'PY'
import sys, statistics, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 3: rows[p[0]].append(float(p[1]))
base = statistics.median(rows["A=1"])
for k, v in rows.items(): print(f"{k:34s} median {statistics.median(v):8.1f}  ({(statistics.median(v) / base - 1) * 100:+.1f} %)  runs {v}")
PY
