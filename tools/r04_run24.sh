#!/bin/bash
# the unchanged loop, no harness round trips: plain / after install (lazy filter on and off) -- and the stack tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -1 | tee gpurun_out/r04w_install_loop.txt
LR_NO_LAZY_FILTER=1 ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -1 | tee -a gpurun_out/r04w_install_loop.txt
ITERS=80 timeout 600 python tools/ref_loop_ab.py 2>&1 | grep -v Warning | tail -1 | tee -a gpurun_out/r04w_install_loop.txt
timeout 900 python -m pytest tests/test_gpu_reference_stack.py tests/test_gpu_training.py -q -m gpu -x 2>&1 | tail -3
