#!/bin/bash
# Quick A/B of the headline under variations:  bash tools/quick_matrix.sh <tag>
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['roofline']['stage_ms_per_view']
print('$name', d['value'], 'host', d['config']['host_issue_ms_per_step'], {k:v for k,v in s.items() if k.startswith('bin') or k in ('tile_sort','compact')})"; }
EXTRA="" one base X=1
EXTRA="--streams 2" one streams2 X=1
EXTRA="--streams 4" one streams4 X=1
EXTRA="" one pad8k LR_BWD_LDS_PAD=8192
EXTRA="" one pad16k LR_BWD_LDS_PAD=16384
EXTRA="--workload c3box" one c3box X=1
EXTRA="--workload c4shape" one c4shape X=1
EXTRA="--workload c5shape" one c5shape X=1
