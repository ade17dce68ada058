#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py tests/test_gpu_raw.py -m gpu -x -q > gpurun_out/r03s_tests.log 2>&1
tail -5 gpurun_out/r03s_tests.log
timeout 900 python tools/ab_bench.py --knob walk_own --values 0,6,12,20,64 --workloads c3,c3box --rounds 3 --stages preprocess,bin_count,bin_scatter,tile_sort --out gpurun_out/r03s_ab_walk_own.json > gpurun_out/r03s_ab.log 2>&1
tail -30 gpurun_out/r03s_ab.log
