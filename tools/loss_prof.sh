# kernel times of the fused loss (rocprofv3 --stats) at a given resolution:  bash tools/loss_prof.sh [WxH]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
RES=${1:-1920x1080}
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loss -o loss -- python $R/tools/loss_bench.py --resolution $RES --reps 20 2>/dev/null | tail -1
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_loss/loss_results.db $R/gpurun_out/loss_stats.md "loss_bench $RES" > /dev/null
grep -E "k_ssim|k_loss" $R/gpurun_out/loss_stats.md
