#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python tools/ab_bench.py --knob part_scan --values 1,0 --workloads c4shape,c2,c3 --rounds 5 --stages bin_count,bin_scan,bin_scatter --out gpurun_out/r04u_ab_part_scan2.json 2>&1 | grep "part_scan=\|==" | cut -c1-230
