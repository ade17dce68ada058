#!/bin/bash
# PMC passes for the bench workload (one counter group per rocprofv3 run, --kernel-trace only, as the
# MI355X guide prescribes).  Usage (on the GPU box):  bash tools/pmc_run.sh <tag> [bench args...]
# Results: gpurun_out/pmc_<tag>/{sq,fetch,write}/...csv
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_$TAG
ARGS="--views 4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras $*"
for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -o $name -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_$TAG/$name.json 2> $R/gpurun_out/pmc_$TAG/$name.err || echo "pass $name failed"
done
ls -R $R/gpurun_out/pmc_$TAG | head -30
