#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_training.py tests/test_gpu_densify.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/loop_segments.py --hand 2>&1 | grep -v Warning | tail -1 | cut -c1-300
