R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
python examples/train_loop.py --gaussians 2000000 --resolution ${RES:-1920x1080} --iters 200 --log 100 --densify-from 100000 2>&1 | tail -4
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/examples/train_loop.py --gaussians 2000000 --resolution ${RES:-1920x1080} --iters 200 --log 100 --densify-from 100000 2>&1 | tail -2
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_train/train_results.db $R/gpurun_out/train_stats.md "train_loop 2M(->1M trained) 1080p 200 iters"
head -30 $R/gpurun_out/train_stats.md
