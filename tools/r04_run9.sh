#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
ITERS=60 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | tail -2
LR_NO_FWD_LOG=1 ITERS=60 timeout 600 python tools/ref_loop_ab.py --install 2>&1 | tail -2
ITERS=60 timeout 600 python tools/ref_loop_ab.py 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_ref_selfcal.py 2>&1 | tail -6
