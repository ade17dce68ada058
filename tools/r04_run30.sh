#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
echo "--- a full collection at the end of the set-up (tests/ref_loop.resident), nothing frozen" | tee -a gpurun_out/r04q_loop_drift.txt
timeout 600 python tools/loop_drift.py 2>&1 | grep "^pass" | cut -c1-200 | tee -a gpurun_out/r04q_loop_drift.txt
ITERS=80 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | grep -v Warning | tail -1 | tee gpurun_out/r04q_install_loop.txt
ITERS=80 timeout 600 python tools/ref_loop_ab.py 2>&1 | grep -v Warning | tail -1 | tee -a gpurun_out/r04q_install_loop.txt
