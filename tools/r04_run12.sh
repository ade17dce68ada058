#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_variants.py tests/test_gpu_full.py tests/test_gpu_fuzz.py tests/test_gpu_raw.py tests/test_gpu_ref_parity.py tests/test_gpu_reference_on_device.py tests/test_gpu_ref_selfcal.py -q -m gpu 2>&1 | tail -6
timeout 900 python tools/ab_bench.py --knob blend_quad --values 0,2 --workloads c3,c3box,c5shape --rounds 3 --out gpurun_out/r04k_ab_one_select.json 2>&1 | grep "blend_quad=\|==" | cut -c1-240
