#!/bin/bash
# experiment: k_gauss_bwd occupancy sweep (rebuilds gauss_bwd.hip on the GPU box)
cd /root/repo
for wv in 2 3 4; do
  touch luciddreamer_amd/csrc/gauss_bwd.hip
  LR_EXTRA_HIPCC_FLAGS="-DLR_GB_WAVES=$wv" python -m luciddreamer_amd.build >/dev/null 2>&1 || echo BUILD FAIL
  for w in c3 c3box; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $wv', d['config']['workload'][:8], d['value'], 'gauss_bwd', d['roofline']['stage_ms_per_view']['gauss_bwd'], 'compact', d['roofline']['stage_ms_per_view']['compact'])"
  done
done
