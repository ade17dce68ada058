#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/ref_loop_ab.py > gpurun_out/r03p_ref_loop_ab.log 2>&1
tail -12 gpurun_out/r03p_ref_loop_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_raw.py tests/test_gpu_training.py tests/test_gpu_reference_stack.py -m gpu -x -q > gpurun_out/r03p_tests.log 2>&1
tail -4 gpurun_out/r03p_tests.log
