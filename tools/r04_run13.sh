#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python tools/ab_bench.py --knob blend_quad --values 0,1 --workloads c3 --rounds 1 --steps 2 --out gpurun_out/tmp_quad.json 2>&1 | grep "blend_quad=" | cut -c1-60,150-330
timeout 600 python tools/ab_bench.py --knob blend_quad --values 0,1 --also tile_map=1 --workloads c3 --rounds 1 --steps 2 --out gpurun_out/tmp_quad.json 2>&1 | grep "blend_quad=" | cut -c1-60,150-330
timeout 600 python tools/ab_bench.py --knob blend_quad --values 0,1 --also gauss_bwd=0 --workloads c3 --rounds 1 --steps 2 --out gpurun_out/tmp_quad.json 2>&1 | grep "blend_quad=" | cut -c1-60,150-330
