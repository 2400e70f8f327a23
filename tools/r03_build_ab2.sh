#!/bin/bash
# compile-time knobs on another workload: BARGS="--workload c3" bash tools/r03_build_ab2.sh tag "<EXTRA>"...
tag=${1:-r03bb}; shift; mkdir -p gpurun_out/$tag
for rep in 1 2 3; do
  for ex in "$@"; do
    make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
    for i in 1 2; do python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-exact $BARGS 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$BARGS $ex]', j['value'], j['ms_per_step'])"; done
  done
done 2>&1 | tee gpurun_out/$tag/ab.txt
python3 - gpurun_out/$tag/ab.txt <<'PY'
import sys,collections,statistics,re
d=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.match(r"\[(.*)\] ([\d.]+) ([\d.]+)",l)
    if m: d.setdefault(m.group(1),[]).append(float(m.group(2)))
for k,v in d.items(): print(f"[{k:60s}] n={len(v)} median {statistics.median(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}")
PY
make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 2>&1 | grep -E " error"
