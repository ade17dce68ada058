#!/bin/bash
# default bench.py run with the live byte-counter passes; the ViewStreams recovery tests after the header-entry fix
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 600 python bench.py > gpurun_out/r04w_bench_default.json 2> gpurun_out/r04w_bench_default.err ) 2>&1 | tail -4
tail -c 400 gpurun_out/r04w_bench_default.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04w_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], r['frac'], r['traffic'], r['traffic_source'])
print(r['traffic_committed_file'], r['live_pmc'])
print(r['counter_vs_algorithmic_bytes'])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loss.py tests/test_gpu_reference_stack.py -q -m gpu -x 2>&1 | tail -3
