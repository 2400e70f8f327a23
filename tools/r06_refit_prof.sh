R=$PWD
for v in ${LIBS:-woop cur}; do
  export LUMEN_MI_LIBRARY=$R/lumenrenderer_amd/ab/liblumen_mi_$v.so
  rm -rf gpurun_out/rp
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp -- python3 $R/tools/refit.py > $R/gpurun_out/rp.log 2>&1)
  f=$(find gpurun_out/rp -name "*kernel_stats.csv" | head -1)
  echo "#### $v"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'refit' in r["Name"] or 'build_top' in r["Name"] or 'copyBuffer' in r["Name"] or 'fillBuffer' in r["Name"]:
        print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f}')
PY
done
