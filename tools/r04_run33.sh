#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_adam -o adam -- python $R/tools/loop_segments.py --hand > /dev/null 2>&1
db=$(ls $R/gpurun_out/prof_adam/*.db 2>/dev/null | head -1)
python $R/tools/rocprof_summary.py $db $R/gpurun_out/r04q_loop_kernels_hand_wide_adam.md "rocprofv3 --kernel-trace --stats -- python tools/loop_segments.py --hand" > /dev/null
rm -rf $R/gpurun_out/prof_adam
sed -n 7,13p $R/gpurun_out/r04q_loop_kernels_hand_wide_adam.md | cut -c1-140
