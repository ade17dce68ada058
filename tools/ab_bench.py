#!/usr/bin/env python
"""A/B measurement of kernel variants inside ONE process on ONE box (lr_tune_set switches them at run time).

    python tools/ab_bench.py --knob blend_quad --values=-1,0,2 [--workloads c3,c3box] [--rounds 3] [--steps 5] [--also fwd_pair=0]
    tools/diag_env.sh python tools/ab_bench.py --knob bwd_red --values=-1,2 --also blend_quad=2      # a RETIRED kernel as partner:
                                                                   # diagnostics build (python -m luciddreamer_amd.build --diagnostics)

For every workload and every value of the knob, `rounds` alternated measurements of
  * the single-stream per-stage times (HIP events inside the library, bench.py's roofline leg), and
  * the 3-stream headline throughput (views/s),
plus the largest difference of the flat gradient bucket against the first value (same sums, different association: the
variants must agree to float rounding).  Writes gpurun_out/ab_<knob>.json and prints a table.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", required=True)
    ap.add_argument("--values", required=True)
    ap.add_argument("--workloads", default="c3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--stages", default="render_bwd,render_fwd,preprocess,gauss_bwd")
    ap.add_argument("--out", default=None)
    ap.add_argument("--also", default="", help="other knobs held fixed during the run: name=value[,name=value]")
    a = ap.parse_args()
    from luciddreamer_amd import _lib
    for kv in filter(None, a.also.split(",")):
        _lib.tune_set(kv.split("=")[0], int(kv.split("=")[1]))
    values = [int(v) for v in a.values.split(",")]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    args = argparse.Namespace(sh_degree=3, no_fused_accumulate=False, gaussians=None)
    report = {}
    for name in a.workloads.split(","):
        wl = bench.Workload(name, args, 0, 1, dev)
        res = {v: {"views_per_s": [], "stage_ms": []} for v in values}
        ref_flat, diffs = None, {}
        for v in values:                                        # correctness first: one step per value
            _lib.tune_set(a.knob, v)
            step = wl.make_step("views", False, 1)
            step()
            torch.cuda.synchronize()
            wl.finish()
            flat = wl.grads.flat.detach().clone()
            m2d = wl.m2d_grad.detach().clone()
            if ref_flat is None:
                ref_flat, ref_m2d = flat, m2d
            scale = float(ref_flat.abs().max())
            diffs[v] = {"bucket_max_abs_diff_rel": float((flat - ref_flat).abs().max()) / scale,
                        "means2D_max_abs_diff_rel": float((m2d - ref_m2d).abs().max()) / max(float(ref_m2d.abs().max()), 1e-30),
                        "finite": bool(torch.isfinite(flat).all())}
        for r in range(a.rounds):
            for v in values:
                _lib.tune_set(a.knob, v)
                vps, ms, host, _ = bench.run_leg(wl, "views", False, a.streams, a.steps, 1, 1, dev)
                step = wl.make_step("views", False, 1)
                step()
                _lib.profile_enable(True)
                bench.timed(step, a.steps, 1, dev)
                st = _lib.profile_read()
                _lib.profile_enable(False)
                wl.finish()
                res[v]["views_per_s"].append(round(vps, 1))
                res[v]["stage_ms"].append({k: round(t[0] / max(t[1], 1), 5) for k, t in st.items()})
        _lib.tune_set(a.knob, -1)
        out = {}
        print(f"== {wl.label}  (R={wl.R_mean:.0f} V={wl.V_mean:.0f})")
        for v in values:
            stage_best = {k: min(s[k] for s in res[v]["stage_ms"]) for k in a.stages.split(",")}
            out[v] = {"views_per_s": res[v]["views_per_s"], "views_per_s_best": max(res[v]["views_per_s"]),
                      "stage_ms_best": stage_best, "vs_first": diffs[v]}
            print(f"  {a.knob}={v}: views/s {res[v]['views_per_s']}  stage ms (best) {stage_best}  diff {diffs[v]}")
        report[name] = out
        del wl
        torch.cuda.empty_cache()
    path = a.out or os.path.join(ROOT, "gpurun_out", f"ab_{a.knob}.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump({"knob": a.knob, "values": values, "rounds": a.rounds, "steps": a.steps, "streams": a.streams,
               "workloads": report}, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
