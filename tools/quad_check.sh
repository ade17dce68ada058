R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for wl in "--workload c3" "--workload c3box" "--workload c2"; do
python bench.py --no-cpu-baseline $wl | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_view']; print(d['value'], s['preprocess'], s['render_fwd'], s['render_bwd'])"
done
