#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03x_tests.log 2>&1
tail -5 gpurun_out/r03x_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r03x_bench_c3.json 2> gpurun_out/r03x_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03x_bench_c3.json').read().strip().splitlines()[-1])
print(d['value'], d.get('sustained'), d['roofline']['stage_ms_per_view'])
e=d.get('entry_points'); e.pop('note',None); print(e)
PY
