#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_drift -o drift -- python $R/tools/loop_drift.py > $R/gpurun_out/r04q_loop_drift_traced.txt 2>&1
grep "^pass" $R/gpurun_out/r04q_loop_drift_traced.txt
db=$(ls $R/gpurun_out/prof_drift/*.db 2>/dev/null | head -1)
python $R/tools/trace_by_pass.py $db 80 2>&1 | tee $R/gpurun_out/r04q_loop_drift_kernels.txt | cut -c1-330
rm -rf $R/gpurun_out/prof_drift
