#!/bin/bash
# full -m gpu suite + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r03i}
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_tests.log 2>&1
tail -5 gpurun_out/${TAG}_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
tail -c 600 gpurun_out/${TAG}_bench_c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_c3.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','sustained','entry_points')}, indent=1))
print({k:v for k,v in d['parity'].items() if k!='grad_err_vs_tensor_max'})
print(d['roofline']['stage_ms_per_view'], d['roofline']['frac'], d['roofline']['path_frac_moved'])
print({k:(v['value'] if 'value' in v else v) for k,v in d['other_workloads'].items()})
print(d['cpu_baseline'].get('c5_train_loop'))
PY
