#!/bin/bash
tag=${1:-r03k3}; mkdir -p gpurun_out/$tag
run() { (export $1; python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-exact $BARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$BARGS $1', j['value'], j['ms_per_step'])"); }
{ for rep in 1 2 3; do for w in c3 c2t c4 c5; do BARGS="--workload $w"; STEPS=5; run A=1; run LUMEN_MI_TAIL_BELOW=120000; done; done
  for rep in 1 2 3; do BARGS="--mode exact"; STEPS=10; run A=1; run LUMEN_MI_TAIL_BELOW=65536; run LUMEN_MI_TAIL_BELOW=120000; done; } 2>&1 | tee gpurun_out/$tag/knobs.txt
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
