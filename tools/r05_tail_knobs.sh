#!/bin/bash
# path-tail threshold / pairing per workload (same build, environment variants, interleaved rounds): WL=<workload> ARGS="<bench args>" bash tools/r05_tail_knobs.sh "<ENV>" ...
mkdir -p gpurun_out/r05; out=gpurun_out/r05/tail_knobs_${WL}.txt; : > $out
for round in 1 2 3; do for v in "$@"; do
  line=$(export $v; timeout 200 python bench.py --workload $WL $ARGS --no-cpu-baseline --no-exact --no-other-reuse 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['rays_per_wave'])")
  echo "$v $line" >> $out
done; done
python3 - $out $WL <<'PY'
import sys, statistics, collections
rows = collections.defaultdict(list); waves = None
for l in open(sys.argv[1]):
    p = l.split(None, 3)
    if len(p) >= 3: rows[p[0]].append(float(p[1])); waves = p[3].strip() if len(p) > 3 else waves
base = statistics.median(rows["A=1"])
print(f"## {sys.argv[2]}  rays per wave {waves}")
for k, v in rows.items(): print(f"{k:52s} median {statistics.median(v):8.1f}  ({(statistics.median(v) / base - 1) * 100:+.1f} %)  runs {v}")
PY
