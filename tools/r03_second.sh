#!/bin/bash
tag=${1:-r03b}; R=$PWD; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -q -s -k "resample_and_combine or contracted_bsdf or srgb_flagged" 2>&1 | grep -v "^$" | cut -c1-900 | head -150 > gpurun_out/$tag/pytest_new.log; cat gpurun_out/$tag/pytest_new.log | grep -v "amdgpu.ids" | tail -30
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; python3 -c "
import json;j=json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'],j['value_exact'],j['device_ms_per_traceframe'])"
