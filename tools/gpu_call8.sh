#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03m_tests.log 2>&1
tail -4 gpurun_out/r03m_tests.log
timeout 900 python tools/ab_bench.py --knob gauss_bwd --values 0,1 --workloads c3,c3box --rounds 2 --steps 4 --stages gauss_bwd --out gpurun_out/r03m_ab_gauss_bwd.json > gpurun_out/r03m_ab.log 2>&1
grep -v "^$" gpurun_out/r03m_ab.log | tail -12
