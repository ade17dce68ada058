#!/bin/bash
# round 4, GPU call 1: the TILE shape of the blend backward (one wave per tile, four pixels per lane) against the others
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_variants.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python tools/ab_bench.py --knob blend_quad --values 0,2,1 --workloads c3,c3box,c4shape,c5shape --rounds 3 --out gpurun_out/r04a_ab_bwd_shape.json 2>&1 | grep -v "^$" | tail -30
