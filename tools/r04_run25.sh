#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for st in "" 1; do
  echo "single-thread backward: '$st'"
  LR_SINGLE_THREAD_BACKWARD=$st timeout 600 python tools/loop_segments.py 2>&1 | grep -v Warning | tail -2 | cut -c1-420
done | tee gpurun_out/r04q_loop_segments_backward_thread.txt
