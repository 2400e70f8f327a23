export LUMEN_MI_FAST_RESAMPLE=1
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['device_ms_per_traceframe'])"; }
run A=1
run LUMEN_MI_TAIL_BELOW=65536
run LUMEN_MI_TAIL_BELOW=131072
run LUMEN_MI_TAIL_BELOW=300000
run LUMEN_MI_TAIL_BELOW=131072 LUMEN_MI_TAIL_LANES=32
run LUMEN_MI_TAIL_BELOW=131072 LUMEN_MI_TAIL_LANES=8
run LUMEN_MI_PICK_AHEAD=0
run LUMEN_MI_SHADOW_ON_WAVE=1
run LUMEN_MI_TAIL_BELOW=131072 LUMEN_MI_SHADOW_ON_WAVE=1
run LUMEN_MI_AUX3_PRIORITY=1
run A=2
