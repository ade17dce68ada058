#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_ref_parity.py -m gpu -x -q > gpurun_out/r03k_gpu_tests.log 2>&1
tail -4 gpurun_out/r03k_gpu_tests.log
timeout 900 python tools/ab_bench.py --knob bwd_red --values 1 --workloads c3,c2,c3box,c4shape,c5shape --rounds 3 --steps 4 --stages tile_sort,bin_scatter,bin_count,compact,bin_scan --out gpurun_out/r03k_sort.json > gpurun_out/r03k_sort.log 2>&1
grep -v "^$" gpurun_out/r03k_sort.log | tail -12
