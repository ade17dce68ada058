#!/bin/bash
# the installed render's visibility filter keeps the loop's masked max-radii update on the device: the unchanged C5 loop after install()
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_stack.py -q -m gpu -x 2>&1 | tail -3
ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -8 | tee gpurun_out/r04w_install_loop.txt
