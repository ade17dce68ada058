#!/usr/bin/env python
"""The UNCHANGED reference training iteration (tests/ref_loop.py = R/luciddreamer.py:283-327) after luciddreamer_amd.install(),
with and without fuse_step (the optimizer step taken by the backward pass), on two scenes at 512 x 512, batch 1:

    c5     1 M Gaussians, box cloud perturbed (BASELINE.json config 5's shape: every Gaussian in view)
    ld512  1 M pixel-sized Gaussians in one layer on a panorama band (LucidDreamer's own statistics: a quarter in view)

    python tools/loop_fuse_step.py [--iters 60] [--scene c5,ld512] [--modes plain,fused] [--passes 3]

Prints ms per iteration (best pass after a warm-up pass).  Under `rocprofv3 --kernel-trace --stats` the same run gives the
kernel table of the loop (profiles/r06n_*)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import luciddreamer_amd                                  # noqa: E402
from luciddreamer_amd import cameras, config, synthetic  # noqa: E402
from tests import ref_loop                                # noqa: E402
from tests.test_gpu_reference_stack import _perturbed, _targets   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--scene", default="c5,ld512")
ap.add_argument("--modes", default="plain,fused")
ap.add_argument("--passes", type=int, default=3)
a = ap.parse_args()
P, W, H = 1_000_000, 512, 512
for scene in a.scene.split(","):
    if scene == "c5":
        cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
        base, hidden = _perturbed(P, 41)
    else:
        cams = cameras.rotate360_path(W, H, n_views=8)
        base = synthetic.make_cloud(P, "shell", 0)
        g = torch.Generator().manual_seed(42)
        hidden = {k: v.clone() for k, v in base.items()}
        hidden["means3D"] = hidden["means3D"] + 0.002 * torch.randn(P, 3, generator=g)
        hidden["shs"] = hidden["shs"] + 0.05 * torch.randn(hidden["shs"].shape, generator=g)
    targets, depths = _targets(hidden, cams)
    order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=a.iters)]
    for mode in a.modes.split(","):
        config.reset()
        config.set_async(True, on_overflow=os.environ.get("LR_POLICY", "verify"))
        best, vis = None, None
        for p in range(a.passes + 1):
            with ref_loop.stack("ours") as (R, dev):
                h = luciddreamer_amd.install(R, backward_on_calling_thread=True, fuse_step=(mode == "fused"))
                try:
                    gm = ref_loop.model_from_cloud(R, base, dev)
                    cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, a.iters)
                    seen = []
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    res = ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=a.iters, opt=opt_r,
                                         on_iteration=(lambda it, gm_, pkg, loss: seen.append(pkg["radii"])) if p == 0 else None)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / a.iters * 1e3
                    if p == 0:
                        vis = float(torch.stack([(r > 0).float().mean() for r in seen[:8]]).mean())
                    else:
                        best = dt if best is None else min(best, dt)
                finally:
                    luciddreamer_amd.uninstall(h)
        print(f"{scene:6s} {mode:6s} {best:7.3f} ms / iteration   (visible fraction {vis:.2f}, final loss {float(res['loss'][-1]):.5f})",
              flush=True)
