#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/ab_bench.py --knob bwd_red --values 1,2 --workloads c3,c3box --rounds 3 --steps 4 --stages render_bwd,render_fwd --out gpurun_out/r03o_extra.json > gpurun_out/r03o_extra.log 2>&1
grep -v "^$" gpurun_out/r03o_extra.log | tail -3
