#!/bin/bash
# full -m gpu suite + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03e_gpu_tests.log 2>&1
tail -5 gpurun_out/r03e_gpu_tests.log
true
