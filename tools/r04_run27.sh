#!/bin/bash
# the count kernel reserves its ranges with atomics (no k_part_scan1): tests, then A/B against the scan kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_shapes.py tests/test_gpu_full.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python tools/ab_bench.py --knob part_scan --values 1,0 --workloads c3,c3box,c5shape,c2 --rounds 3 --stages bin_count,bin_scan,bin_scatter,tile_sort --out gpurun_out/r04u_ab_part_scan.json 2>&1 | grep "part_scan=\|==" | cut -c1-260
