#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/host_floor.py 2>&1 | grep -v Warning | tee gpurun_out/r04w_host_floor.txt
