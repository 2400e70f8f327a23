#!/bin/bash
# the GPU suite exactly as the driver runs it (-x -q -m gpu), with durations; then the numbers the round's notes quote
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -45 > gpurun_out/r05/pytest_gpu.log; cat gpurun_out/r05/pytest_gpu.log
timeout 600 python tools/tex_filter_gap.py > gpurun_out/r05/tex_filter_gap.txt 2>&1; cat gpurun_out/r05/tex_filter_gap.txt
timeout 600 python tools/fp16_blend_chain.py > gpurun_out/r05/fp16_blend_chain.txt 2>&1; cat gpurun_out/r05/fp16_blend_chain.txt
