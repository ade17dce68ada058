#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_variants.py tests/test_gpu_full.py tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -4
timeout 900 python tools/ab_bench.py --knob bwd_red --values 4,1 --also blend_quad=0 --workloads c3,c3box,c5shape --rounds 3 --out gpurun_out/r04i_ab_bwd_nocheck_half.json 2>&1 | grep "bwd_red=\|==" | cut -c1-230
timeout 900 python tools/ab_bench.py --knob bwd_red --values 4,1 --also blend_quad=2 --workloads c3,c3box --rounds 3 --out gpurun_out/r04i_ab_bwd_nocheck_tile.json 2>&1 | grep "bwd_red=\|==" | cut -c1-230
