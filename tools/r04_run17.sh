#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_gpu_fuzz.py tests/test_gpu_shapes.py tests/test_gpu_raw.py -q -m gpu 2>&1 | tail -4
timeout 900 python tools/ab_bench.py --knob bwd_red --values 4,1 --workloads c3,c3box,c2 --rounds 3 --out gpurun_out/r04m_ab_easy_chunks.json 2>&1 | grep "bwd_red=\|==" | cut -c1-250
