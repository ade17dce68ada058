#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_reference_stack.py::test_install_switches_the_unchanged_loop_onto_the_fused_pieces" -q -m gpu 2>&1 | tail -6
timeout 600 python tools/host_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_host_breakdown.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04f_bench_c3.json 2> gpurun_out/r04f_bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f_bench_c3.json').read().strip().splitlines()[-1])
print(d['value'], d['sustained'], {k:v for k,v in d['entry_points'].items() if k!='note'})
PY
