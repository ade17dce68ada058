#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_raw.py tests/test_gpu_variants.py tests/test_cabi_and_api.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -6
timeout 600 python tools/host_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "ViewStreams(3\|lr_views" | tee gpurun_out/r04g_host_breakdown.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04g_bench_c3.json 2> gpurun_out/r04g_bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04g_bench_c3.json').read().strip().splitlines()[-1])
print(d['value'], d['sustained'], {k:v for k,v in d['entry_points'].items() if k!='note'})
PY
