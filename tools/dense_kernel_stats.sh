#!/bin/bash
# rocprofv3 kernel-trace summaries of the dense workloads (single stream):  bash tools/dense_kernel_stats.sh <tag>
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for wl in c3box c5shape c4shape; do
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$wl -o ${TAG}_$wl -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extras --steps 4 --warmup 2 --sustain-seconds 0 --streams 1 > /dev/null 2>&1
  cd $R
  db=$(ls gpurun_out/prof_${TAG}_$wl/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_${wl}_kernel_stats_streams1.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-extras --steps 4 --warmup 2 --sustain-seconds 0 --streams 1"
  rm -rf gpurun_out/prof_${TAG}_$wl
  head -16 gpurun_out/${TAG}_${wl}_kernel_stats_streams1.md | tail -9 | cut -c1-130
done
