R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_gpu_fuzz.py tests/test_gpu_raw.py -x -q -m gpu 2>&1 | tail -3
for wl in "--workload c3" "--workload c3box" "--workload c3"; do
python bench.py --no-cpu-baseline $wl | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_view']; print(d['value'], s['preprocess'], s['render_fwd'], s['render_bwd'], s['gauss_bwd'])"
done
