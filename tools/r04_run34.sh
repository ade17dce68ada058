#!/bin/bash
# forward blend at 23 instructions per step: parity first, evidence only if green
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_gpu_variants.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_ref_parity.py tests/test_gpu_ref_selfcal.py tests/test_gpu_reference_on_device.py tests/test_gpu_raw.py -q -m gpu -x > gpurun_out/r04n_gputests.log 2>&1
tail -3 gpurun_out/r04n_gputests.log
grep -q " passed" gpurun_out/r04n_gputests.log && ! grep -q "failed\|error" gpurun_out/r04n_gputests.log || exit 1
timeout 900 bash tools/gpu_round.sh r04n 2>&1 | tail -2 | cut -c1-400
