#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_tests.log 2>&1
tail -3 gpurun_out/r04a_tests.log
for wl in c3 c5shape c3box; do
timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r04a_bench_$wl.json 2> gpurun_out/r04a_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04a_bench_$wl.json').read().strip().splitlines()[-1])
print('$wl', d['value'], d['roofline']['stage_ms_per_view'])
PY
done
