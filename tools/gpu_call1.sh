#!/bin/bash
# round-3 call 1: instruction-cost microbenchmark + reduction variants of the blend backward
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/ab_bench.py --knob bwd_red --values 11,111,1,101,1001,1101 --workloads c3,c3box,c5shape --rounds 3 --steps 4 > gpurun_out/r03d_ab_bwd_red.log 2>&1
tail -30 gpurun_out/r03d_ab_bwd_red.log
