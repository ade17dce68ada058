#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03f}
bash tools/pmc_run.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}_summary.json "tools/pmc_run.sh $TAG"
python - <<PY
import json
d=json.load(open('gpurun_out/pmc_${TAG}_summary.json'))
print(d.get('_lr_version'))
for k,v in d.items():
    if isinstance(v,dict):
        print(k, {c:round(x) for c,x in v.items() if c in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','FETCH_SIZE','WRITE_SIZE','GRBM_GUI_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_WAIT_INST_ANY')})
PY
