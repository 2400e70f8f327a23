#!/bin/bash
# after the tile-copy change: the rehearsal loop again, then the new tests, then the filter-gap numbers
bash tools/r05_c4_repro.sh fixed_c4 14 c4 8
bash tools/r05_c4_repro.sh fixed_c2 8 c2 8
cat gpurun_out/r05_c4_repro/fixed_*/summary.txt > gpurun_out/r05_fixed_summary.txt
timeout 1500 python -m pytest tests/test_gpu_lowpoly.py tests/test_zz_multiprocess.py "tests/test_gpu_parity.py::test_device_texture_fetch_follows_the_published_cuda_filter_rule" "tests/test_gpu_parity.py::test_textured_materials_match_oracle" "tests/test_gpu_parity.py::test_c2_textured_at_full_size_is_bit_exact_against_the_oracle" "tests/test_gpu_parity.py::test_stitched_tiles_equal_the_single_gpu_frame" -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r05_newtests.log; tail -30 gpurun_out/r05_newtests.log
timeout 600 python tools/tex_filter_gap.py > gpurun_out/r05_tex_filter_gap.txt 2>&1; cat gpurun_out/r05_tex_filter_gap.txt
timeout 600 python bench.py --workload lowpoly --steps 32 --warmup 8 > gpurun_out/r05_bench_lowpoly.json 2> gpurun_out/r05_bench_lowpoly.err; tail -c 1500 gpurun_out/r05_bench_lowpoly.json
