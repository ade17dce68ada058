#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_round.sh r04z 2>&1 | tail -8
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z_bench_c3.json').read().strip().splitlines()[-1])
p=d['parity']
print('parity', {k:v for k,v in p.items() if k not in ('per_view','reference_self_disagreement','grad_err_vs_tensor_max')})
print('grad', p['grad_err_vs_tensor_max'])
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('stage_ms_per_view',)})
print('c5', d['cpu_baseline'].get('c5_train_loop'))
PY
