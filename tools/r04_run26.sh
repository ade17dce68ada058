#!/bin/bash
# kernel time of the installed unchanged loop vs the hand-edited one (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
for v in install hand; do
  arg=""; [ $v = hand ] && arg="--hand"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loop_$v -o loop_$v -- python $R/tools/loop_segments.py $arg > /dev/null 2>&1
  db=$(ls $R/gpurun_out/prof_loop_$v/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db $R/gpurun_out/r04w_loop_kernels_$v.md "rocprofv3 --kernel-trace --stats -- python tools/loop_segments.py $arg (3 passes x 60 iterations)"
  rm -rf $R/gpurun_out/prof_loop_$v
  head -40 $R/gpurun_out/r04w_loop_kernels_$v.md | cut -c1-150
done
