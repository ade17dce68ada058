#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/loop_drift.py --profile 2>&1 | grep -v Warning | tail -50 | tee gpurun_out/r04q_loop_drift.txt
