#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/r03l_dist.log 2>&1
tail -15 gpurun_out/r03l_dist.log
