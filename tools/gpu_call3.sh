#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VALS=${1:-0,1}
LR_TUNE=${2:-} timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_parity.py tests/test_gpu_raw.py tests/test_gpu_ref_parity.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03g_gpu_tests.log 2>&1
tail -4 gpurun_out/r03g_gpu_tests.log
timeout 900 python tools/ab_bench.py --knob preprocess --values $VALS --workloads c3,c3box --rounds 3 --steps 4 --stages preprocess --out gpurun_out/r03g_ab_preprocess.json > gpurun_out/r03g_ab_preprocess.log 2>&1
grep -v "^$" gpurun_out/r03g_ab_preprocess.log | tail -30
