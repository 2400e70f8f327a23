#!/bin/bash
# round 4: the Sandbox's own workload — full-size oracle tests in both arithmetic modes, then the bench line (profiles/r04_bench_sandbox.json)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "sandbox_default or 1440p_odd_depth" 2>&1 | tail -15
python bench.py --workload sandbox --steps 32 --warmup 8 > gpurun_out/r04_bench_sandbox.json 2> gpurun_out/r04_bench_sandbox.err
tail -c 600 gpurun_out/r04_bench_sandbox.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04_bench_sandbox.json"))
print({k: j[k] for k in ("value", "ms_per_step", "value_exact", "ms_per_step_exact", "device_ms_per_traceframe")})
print(j["config"]["rays_per_wave"], j["cpu_baseline"])
PY
