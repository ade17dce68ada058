#!/bin/bash
# micro-optimised blend backward (fma chain for the colour difference, one LDS address per candidate): tests + evidence
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04s_gputests.log 2>&1
tail -5 gpurun_out/r04s_gputests.log
timeout 1500 bash tools/gpu_round.sh r04s 2>&1 | tail -40
