#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
LR_POLICY=drop timeout 600 python tools/loop_segments.py 2>&1 | grep -v Warning | tail -2 | tee gpurun_out/r04w_loop_segments_drop.txt
