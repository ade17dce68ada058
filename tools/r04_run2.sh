#!/bin/bash
# round 4, GPU call 2: TILE shape with all waves resident (8 per SIMD) vs 7 per SIMD vs the 2-wave shape; where QUAD differs;
# the reference's disagreement with itself (parity budgets); VALU instruction counts of the shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_shapes.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/shape_diff.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python tools/ab_bench.py --knob blend_quad --values 0,2 --workloads c3,c3box,c4shape --rounds 3 --out gpurun_out/r04b_ab_bwd_tile8.json 2>&1 | grep "blend_quad=\|==" | cut -c1-260
timeout 900 python tools/ab_bench.py --knob blend_quad --values 0,2 --also bwd_red=3 --workloads c3,c3box --rounds 2 --out gpurun_out/r04b_ab_bwd_tile7.json 2>&1 | grep "blend_quad=\|==" | cut -c1-260
timeout 1500 python -m pytest tests/test_gpu_ref_selfcal.py -x -q -m gpu -s 2>&1 | grep -v "^$" | cut -c1-1500 | tail -12
cd /tmp
for shape in 0 2; do
  LR_BLEND_QUAD_BWD=$shape rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_r04b_shape$shape -o sq -- python $R/bench.py --views 4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_r04b_shape$shape/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "render_bwd" not in k and "render_fwd" not in k: continue
    k = k.split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in acc:
    print("shape $shape", k, {c: round(v / n[k]) for c, v in acc[k].items()}, "launches", n[k])
PY
  rm -rf $R/gpurun_out/pmc_r04b_shape$shape
done
