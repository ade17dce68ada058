#!/bin/bash
# Round-end evidence on the GPU box:  bash tools/gpu_round.sh <tag>
#   bench.py (driver settings) -> gpurun_out/<tag>_bench_c3.json
#   rocprofv3 --kernel-trace --stats of the same command (default streams and --streams 1) -> <tag>_c3_kernel_stats_*.md
#   PMC passes (tools/pmc_run.sh) -> <tag>_pmc_c3.json, stamped with lr_version()
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
tail -c 300 gpurun_out/${TAG}_bench_c3.err
for wl in c2 c3box c4shape c5shape ld512; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_${wl}.json 2>/dev/null
done
# single-stream kernel statistics of the 512^2 dense shape (one view alone on the GPU: what the training loop sees)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_c5shape -o ${TAG}_c5shape -- python $R/bench.py --workload c5shape --no-cpu-baseline --no-extras --steps 5 --warmup 2 --sustain-seconds 0 --streams 1 > /dev/null 2>&1)
db=$(ls gpurun_out/prof_${TAG}_c5shape/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_c5shape_kernel_stats_streams1.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload c5shape --no-cpu-baseline --no-extras --steps 5 --warmup 2 --sustain-seconds 0 --streams 1"
rm -rf gpurun_out/prof_${TAG}_c5shape
cd /tmp
CMD="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --sustain-seconds 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_default -o ${TAG}_default -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_streams1 -o ${TAG}_streams1 -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --sustain-seconds 0 --streams 1 > /dev/null 2>&1
cd $R
for v in default streams1; do
  db=$(ls gpurun_out/prof_${TAG}_$v/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_c3_kernel_stats_$v.md "rocprofv3 --kernel-trace --stats -- $CMD$([ $v = streams1 ] && echo ' --streams 1')"
done
bash tools/pmc_run.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc_c3.json "tools/pmc_run.sh $TAG"
rm -rf gpurun_out/pmc_$TAG/*/*.csv gpurun_out/pmc_$TAG/*/*.db
# the shapes LucidDreamer itself runs (512^2, every Gaussian in view) and the dense 1080p cloud, same counters
for wl in c5shape c3box; do
  bash tools/pmc_run.sh ${TAG}_$wl --workload $wl > gpurun_out/pmc_${TAG}_$wl.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$wl gpurun_out/${TAG}_pmc_$wl.json "tools/pmc_run.sh ${TAG}_$wl --workload $wl"
  rm -rf gpurun_out/pmc_${TAG}_$wl/*/*.csv gpurun_out/pmc_${TAG}_$wl/*/*.db
done
rm -rf gpurun_out/prof_${TAG}_default gpurun_out/prof_${TAG}_streams1
head -30 gpurun_out/${TAG}_c3_kernel_stats_streams1.md
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_c3.json').read().strip().splitlines()[-1])
print(d['value'], d['sustained'], d['entry_points'])
print(d['roofline']['stage_ms_per_view'])
PY
