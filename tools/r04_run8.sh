#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -12
timeout 300 python -m pytest tests/test_gpu_ref_parity.py -q -m gpu -s -k "needles" 2>&1 | grep "needles vs" 
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/gpu_round.sh r04h 2>&1 | tail -45
