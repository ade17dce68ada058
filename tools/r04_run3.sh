#!/bin/bash
# round 4, GPU call 3: tests of this round's host-side work (ViewStreams recovery, install(), calibration with the
# default-contraction reference build, shapes without conditional skips) and a full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_loss.py tests/test_gpu_training.py "tests/test_gpu_reference_stack.py::test_install_switches_the_unchanged_loop_onto_the_fused_pieces" tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_ref_selfcal.py -q -m gpu -s 2>&1 | grep -v "^{" | tail -12
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04c_bench_c3.json 2> gpurun_out/r04c_bench_c3.err
tail -c 400 gpurun_out/r04c_bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c_bench_c3.json').read().strip().splitlines()[-1])
print(d['value'], d['sustained'], d['entry_points'])
print(d['roofline']['stage_ms_per_view'])
p=d['parity']; print({k:v for k,v in p.items() if k not in ('per_view','reference_self_disagreement')})
print(d['cpu_baseline'].get('c5_train_loop'))
PY
