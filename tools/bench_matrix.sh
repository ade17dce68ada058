#!/bin/bash
# Bench lines for the other BASELINE.json shapes and API variants:  bash tools/bench_matrix.sh <tag>
# (C3 default is tools/profile_round.sh).  One JSON line per file under gpurun_out/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench_${name}.json 2> gpurun_out/${TAG}_bench_${name}.err; python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_bench_${name}.json')); print('${name}', d['value'], d['unit'], d['config']['api'])"; }
run c2 --workload c2
run c3box --workload c3box
run c4shape --workload c3box --gaussians 3000000 --resolution 2560x1440 --views 6
run c5shape --workload c3box --resolution 512x512
run c3_autograd --api autograd
run c3_views_loss --api views-loss
run c3_exact --api autograd --exact
