#!/bin/bash
# Round-end evidence on the GPU box:  bash tools/profile_round.sh <tag>
#   1. bench.py (default C3) -> gpurun_out/<tag>_bench_c3.json
#   2. rocprofv3 --kernel-trace --stats of the same command, default streams and --streams 1
# PMC passes are separate (tools/pmc_run.sh).  Summaries: tools/rocprof_summary.py on the merged .db files.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
tail -c 600 gpurun_out/${TAG}_bench_c3.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_default -o ${TAG}_default -- python $R/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_streams1 -o ${TAG}_streams1 -- python $R/bench.py --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
ls $R/gpurun_out/prof_${TAG}_default $R/gpurun_out/prof_${TAG}_streams1
