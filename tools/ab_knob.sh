#!/bin/bash
# Interleaved A/B of tuning-key settings on ONE box (round 4's single driver for knob experiments; replaces the r03_knobs*.sh family).
#   bash tools/ab_knob.sh OUT.txt ROUNDS "BENCH ARGS" "ENV_A" "ENV_B" ...
# Every variant is a set of LUMEN_MI_* environment overrides ("-" = none); ROUNDS rounds, each runs every variant once in turn, so that drift of the
# box (clocks, neighbours) hits all variants alike.  Prints per variant the median / min / max of bench.py's value (Mrays/s) and of value_lazy_reuse.
out=$1; rounds=$2; args=$3; shift 3
mkdir -p gpurun_out
: > gpurun_out/ab_raw.txt
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    line=$(env $envs python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1)
    echo "$v|$line" >> gpurun_out/ab_raw.txt
  done
done
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
    f.write(f"# bench.py --no-cpu-baseline {args}; variants interleaved on one box, {max(len(r) for r in rows.values())} rounds\n")
    for v, js in rows.items():
        for key in ("value", "value_lazy_reuse", "value_exact"):
            xs = [j[key] for j in js if j.get(key) is not None]
            if xs:
                f.write(f"[{v:48s}] {key:18s} n={len(xs)} median {statistics.median(xs):9.1f} min {min(xs):9.1f} max {max(xs):9.1f}\n")
        tails = [j["device_ms_per_traceframe"].get("tail") for j in js]
        f.write(f"[{v:48s}] device ms per TraceFrame, tail class: {[round(t, 3) for t in tails if t is not None]}\n")
print(open(out).read())
PY
