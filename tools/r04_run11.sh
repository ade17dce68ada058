#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_shapes.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -4
timeout 900 python tools/ab_bench.py --knob blend_quad --values 0,2 --workloads c3,c3box,c4shape --rounds 4 --out gpurun_out/r04j_ab_half_checkfree_vs_tile.json 2>&1 | grep "blend_quad=\|==" | cut -c1-240
