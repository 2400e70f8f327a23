#!/bin/bash
# schedule knobs under the fast ReSTIR mode (bench default): one bench line per environment setting
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['device_ms_per_traceframe'])"; }
run A=1
run LUMEN_MI_AUX_PRIORITY=0
run LUMEN_MI_AUX_PRIORITY=0 LUMEN_MI_AUX3_PRIORITY=1
run LUMEN_MI_AUX3_PRIORITY=1
run LUMEN_MI_TAIL_BELOW=100000
run LUMEN_MI_TAIL_BELOW=32768
run LUMEN_MI_PICK_AHEAD=0
run A=2
