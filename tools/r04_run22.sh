#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/loop_profile.py 2>&1 | grep -v Warning | head -60 | tee gpurun_out/r04w_loop_profile.txt
python - <<PY
from luciddreamer_amd import dropin
print("lazy assignments counted in this process:", dropin.lazy_assignments)
PY
