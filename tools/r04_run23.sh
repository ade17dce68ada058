#!/bin/bash
# the unchanged loop with the harness's own setup outside the clock: install variants, and a host profile of one
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -4 | tee gpurun_out/r04w_install_loop.txt
ITERS=80 timeout 600 python tools/ref_loop_ab.py 2>&1 | grep -v Warning | tail -2 | tee -a gpurun_out/r04w_install_loop.txt
timeout 600 python tools/loop_profile.py 2>&1 | grep -v Warning | head -45 | tee gpurun_out/r04w_loop_profile.txt
