#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ref_parity.py tests/test_gpu_reference_on_device.py -m gpu -q -s -k "ill_conditioned or committed" > gpurun_out/r03l_dist.log 2>&1
tail -40 gpurun_out/r03l_dist.log
