#!/bin/bash
# backward on the calling thread: install() and ViewStreams
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_stack.py tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_distributed.py -q -m gpu -x 2>&1 | tail -3
ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -1 | tee gpurun_out/r04q_install_loop.txt
timeout 600 python tools/host_breakdown.py 2>&1 | grep -v Warning | tail -9 | tee gpurun_out/r04q_host_breakdown.txt
