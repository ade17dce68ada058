#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/shape_vs_oracle.py 15,18,21,27 2>&1 | grep -v amdgpu.ids | cut -c1-300
timeout 600 python tools/shape_diff.py 0,9,15,18,21,27 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_calibration.json'))
for k,v in d['cases'].items():
    x=v['hip_vs_reference_gfx950']; print(k, 'hip vs strict ref: pixels', x['pixels_beyond_1e-5'], 'rows', x['rows_beyond_1e-4'], 'worst', {a:f"{b:.1e}" for a,b in x['worst_row_rel'].items()})
PY
