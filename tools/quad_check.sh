R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
LR_BLEND_QUAD_BWD=1 timeout 600 python -m pytest tests/test_gpu_full.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
LR_BLEND_QUAD_BWD=0 timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_view']; print(d['value'], s['render_fwd'], s['render_bwd'])"
python bench.py --no-cpu-baseline --workload c3box --resolution 512x512| python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_view']; print(d['value'], s['render_fwd'], s['render_bwd'])"
