#!/bin/bash
# Interleaved A/B of COMPILE-TIME variants without building on the GPU box.  Two steps:
#   here (no GPU):   bash tools/ab_lib.sh build NAME "EXTRA flags"       -> lumenrenderer_amd/ab/liblumen_mi_NAME.so   (repeat per variant; "-" = no extra flags)
#   on the box:      bash tools/ab_lib.sh run OUT.txt ROUNDS "BENCH ARGS" NAME_A NAME_B ...
# run: every round runs bench.py once per variant in turn (LUMEN_MI_LIBRARY selects the build); prints median / min / max of value, value_lazy_reuse, value_exact.
mode=$1; shift
if [ "$mode" = build ]; then
  name=$1; ex=$2; [ "$ex" = "-" ] && ex=""
  mkdir -p lumenrenderer_amd/ab
  make -C lumenrenderer_amd/csrc clean > /dev/null; make -C lumenrenderer_amd/csrc -j8 EXTRA="$ex" 2>&1 | grep -E " error|warning: .*spill"
  cp lumenrenderer_amd/liblumen_mi.so lumenrenderer_amd/ab/liblumen_mi_$name.so; echo "built $name ($ex)"
  exit 0
fi
out=$1; rounds=$2; args=$3; shift 3
mkdir -p gpurun_out; : > gpurun_out/ab_raw.txt
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    line=$(LUMEN_MI_LIBRARY=$PWD/lumenrenderer_amd/ab/liblumen_mi_$v.so timeout 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1)
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
    f.write(f"# bench.py --no-cpu-baseline {args}; prebuilt library per variant (LUMEN_MI_LIBRARY), variants interleaved on one box\n")
    for v, js in rows.items():
        for key in ("value", "value_lazy_reuse", "value_exact"):
            xs = [j[key] for j in js if j.get(key) is not None]
            if xs:
                f.write(f"[{v:32s}] {key:18s} n={len(xs)} median {statistics.median(xs):9.1f} min {min(xs):9.1f} max {max(xs):9.1f}\n")
        cl = [(round(j["device_ms_per_traceframe"].get("closest", 0), 3), round(j["device_ms_per_traceframe"].get("shadow", 0), 3), round(j["device_ms_per_traceframe"].get("tail", 0), 3)) for j in js]
        f.write(f"[{v:32s}] device ms per TraceFrame (closest, shadow, tail): {cl}\n")
print(open(out).read())
PY
