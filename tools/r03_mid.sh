#!/bin/bash
tag=${1:-r03m}; mkdir -p gpurun_out/$tag
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/$tag/pytest.log; cat gpurun_out/$tag/pytest.log
bash tools/env_ab.sh $tag "${KREGEX:-temporal}" "A=1" "A=2" 2>&1 | tee gpurun_out/$tag/ab.txt
