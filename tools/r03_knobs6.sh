#!/bin/bash
tag=${1:-r03k6}; mkdir -p gpurun_out/$tag
run() { (export $1 $2; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact $BARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 $2', j['value'], j['ms_per_step'])"); }
python __graft_entry__.py smoke 2>&1 | tail -1
{ for rep in 1 2 3 4 5; do run A=1; run LUMEN_MI_AUX3_PRIORITY=1; run LUMEN_MI_TAIL_BELOW=120000; run LUMEN_MI_TAIL_BELOW=95000; done; } 2>&1 | tee gpurun_out/$tag/knobs.txt
python3 - gpurun_out/$tag/knobs.txt <<'PY'
import sys,collections,statistics
d=collections.OrderedDict()
for l in open(sys.argv[1]):
    p=l.split()
    try: v=float(p[-2])
    except: continue
    d.setdefault(" ".join(p[:-2]),[]).append(v)
for k,v in d.items(): print(f"{k:60s} n={len(v)} median {statistics.median(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}")
PY
