#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_ref_selfcal.py -q -m gpu -s 2>&1 | grep "strict mode\|passed\|failed\|Error\|assert" | cut -c1-600 | tail -14
timeout 600 python tools/ab_bench.py --knob strict --values 0,1 --workloads c3 --rounds 2 --out gpurun_out/r04l_ab_strict.json 2>&1 | grep "strict=\|==" | cut -c1-330
