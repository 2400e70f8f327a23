#!/bin/bash
# usage (GPU box): bash tools/r06_rank_prof.sh OUTDIR "R/N" [bench args] — kernel statistics and steady-state timeline of ONE emulated rank's window (bench.py --emulate-rank), overlapped
out=$1; e=$2; shift 2; R=$PWD; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact --no-other-reuse --emulate-rank $e "$@" > $R/$out/bench.json 2> $R/$out/bench.err)
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].endswith("_inst")][:22]: print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
python3 tools/timeline.py "$t" > $out/timeline.txt; rm -rf $out/prof; tail -1 $out/bench.json | cut -c1-160
