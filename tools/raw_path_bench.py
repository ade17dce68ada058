#!/usr/bin/env python
"""Time one training-style view (forward + backward from the STORED GaussianModel parameters) through
render() (torch activations + cat, then the rasterizer op) and render_raw() (activations inside the kernels).

    python tools/raw_path_bench.py [--gaussians 1000000] [--views 30] [--resolution 1920x1080]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luciddreamer_amd import cameras, config, synthetic                      # noqa: E402
from luciddreamer_amd.gaussian_renderer import GaussianCloud, render, render_raw   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=30)
    ap.add_argument("--resolution", default="1920x1080")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    W, H = (int(v) for v in args.resolution.split("x"))
    dev = torch.device("cuda:0")
    cloud = {k: v.to(dev) for k, v in synthetic.make_cloud(args.gaussians, "band", 0).items()}
    pc = GaussianCloud(cloud["means3D"], cloud["scales"], cloud["rotations"], cloud["opacities"], cloud["shs"])
    cams = [c.to(dev) for c in cameras.rotate360_path(W, H, n_views=args.views)]
    gcol = synthetic.upstream_grad(H, W).to(dev)
    out = {}
    for fused in (False, True):
        config.set_fused_grad_accumulation(fused)
        for name, fn in (("render", render), ("render_raw", render_raw)):
            config.reset()
            for p in pc.parameters():
                p.grad = torch.zeros_like(p) if fused else None
            config.set_async(True, headroom=1.3)
            for c in cams:                                   # first sighting runs exact and sizes the binning capacity
                (fn(c, pc)["render"] * gcol).sum().backward()
            best = 1e9
            for _ in range(args.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for c in cams:
                    (fn(c, pc)["render"] * gcol).sum().backward()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / len(cams))
            config.drain()
            config.set_async(False)
            out[f"{name}{'+fused_accumulate' if fused else ''}"] = round(best * 1e3, 4)
    config.set_fused_grad_accumulation(False)
    print(json.dumps({"workload": f"{args.gaussians} Gaussians, {W}x{H}, rotate360, autograd op, one stream",
                      "ms_per_view": out}))


if __name__ == "__main__":
    main()
