#!/usr/bin/env python
"""Is the kernel shape the library picks the fastest one?  Per workload: the step (views/s) with the library's own rule against
every forced pair (blend backward: 2 waves / 4 waves / 1 wave per tile; blend forward: quadrant / candidate pairs / 1 wave per
tile), with three views in flight (the multi-view entry points) and with one (a lone view: LucidDreamer's own loop).

    python tools/shape_sweep.py [--workloads c2,c3,c3box,c4shape,c5shape,ld512] [--steps 6] [--rounds 2] [--out gpurun_out/shape_sweep.json]

tests/test_gpu_zz_heuristics.py asserts the same on a reduced sweep."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

BWD = {0: "half", 1: "quad", 2: "tile"}
FWD = {0: "quadrant", 1: "pairs", 2: "tile"}


def sweep(name, streams, steps, rounds, dev, combos=None):
    from luciddreamer_amd import _lib
    args = argparse.Namespace(sh_degree=3, no_fused_accumulate=False, gaussians=None, exchange="allreduce")
    wl = bench.Workload(name, args, 0, 1, dev, views=min(bench.WORKLOADS[name][3], 12))
    res = {}
    combos = combos or [(-1, -1)] + [(b, f) for b in (0, 1, 2) for f in (0, 1, 2)]
    for r in range(rounds):
        for b, f in combos:
            _lib.tune_set("blend_quad", b)
            _lib.tune_set("fwd_pair", f)
            try:
                v, ms, host, _ = bench.run_leg(wl, "views", False, streams, steps, 1, 1, dev)
                shapes = _lib.last_launch_shapes()
            finally:
                _lib.tune_set("blend_quad", -1)
                _lib.tune_set("fwd_pair", -1)
            key = "default" if b < 0 else f"bwd {BWD[b]} / fwd {FWD[f]}"
            e = res.setdefault(key, {"views_per_s": [], "shapes": shapes})
            e["views_per_s"].append(round(v, 1))
    del wl
    torch.cuda.empty_cache()
    best = max(max(e["views_per_s"]) for k, e in res.items() if k != "default")
    d = max(res["default"]["views_per_s"])
    return {"default_views_per_s": d, "default_shapes": {"forward": res["default"]["shapes"][0], "backward": res["default"]["shapes"][1]},
            "best_forced_views_per_s": best,
            "best_forced": max((k for k in res if k != "default"), key=lambda k: max(res[k]["views_per_s"])),
            "default_over_best": round(d / best, 4), "all": {k: e["views_per_s"] for k, e in res.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,c3,c3box,c4shape,c5shape,ld512")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shape_sweep.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    report = {}
    for name in a.workloads.split(","):
        report[name] = {}
        for streams in (3, 1):
            r = sweep(name, streams, a.steps, a.rounds, dev)
            report[name][f"{streams}_in_flight"] = r
            print(f"{name:8s} {streams} in flight: default {r['default_shapes']} {r['default_views_per_s']:8.1f} views/s; best forced "
                  f"{r['best_forced']} {r['best_forced_views_per_s']:8.1f}  -> default / best = {r['default_over_best']:.3f}", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
