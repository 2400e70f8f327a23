#!/bin/bash
# Interleaved A/B of COMPILE-TIME variants on one box (round 4's single driver; replaces the r03_build_ab*.sh family).
#   bash tools/ab_build.sh OUT.txt ROUNDS "BENCH ARGS" "EXTRA flags A" "EXTRA flags B" ...      ("-" = no extra flags)
# Each round rebuilds the library with each flag set in turn and runs bench.py twice; prints per variant median / min / max of value and value_lazy_reuse.
out=$1; rounds=$2; args=$3; shift 3
mkdir -p gpurun_out
: > gpurun_out/ab_raw.txt
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then ex=""; else ex="$v"; fi
    make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error"
    for i in 1 2; do
      line=$(python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1)
      echo "$v|$line" >> gpurun_out/ab_raw.txt
    done
  done
done
make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 2>&1 | grep -E " error"
python - "$out" "$args" <<'PY'
import sys, json, statistics
out, args = sys.argv[1], sys.argv[2]
rows = {}
for line in open("gpurun_out/ab_raw.txt"):
    v, _, js = line.partition("|")
    try:
        j = json.loads(js)
    except ValueError:
        continue
    rows.setdefault(v, []).append(j)
with open(out, "a") as f:
    f.write(f"# bench.py --no-cpu-baseline {args}; library rebuilt per variant (make EXTRA=...), variants interleaved on one box\n")
    for v, js in rows.items():
        for key in ("value", "value_lazy_reuse", "value_exact"):
            xs = [j[key] for j in js if j.get(key) is not None]
            if xs:
                f.write(f"[{v:48s}] {key:18s} n={len(xs)} median {statistics.median(xs):9.1f} min {min(xs):9.1f} max {max(xs):9.1f}\n")
        cl = [round(j["device_ms_per_traceframe"].get("closest", 0), 3) for j in js]
        f.write(f"[{v:48s}] device ms per TraceFrame, closest-hit class (packet wave + queue waves): {cl}\n")
print(open(out).read())
PY
