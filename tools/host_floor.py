#!/usr/bin/env python
"""Host time per view of the drop-in path when the GPU is NOT the limit: a tiny scene (2000 Gaussians, 128 x 128), so what is
timed is the Python + binding + launch cost of one view under parallel.ViewStreams, per piece."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import cameras, config, parallel, synthetic  # noqa: E402
from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = torch.device("cuda:0")
P, W, H = 2000, 128, 128
cloud = synthetic.make_cloud(P, "band", 0)
leaf = {k: v.to(dev).requires_grad_(True) for k, v in cloud.items()}
grads = parallel.FlatGrads(list(leaf.values()))
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
m2d.grad = torch.zeros_like(m2d)
g = synthetic.upstream_grad(H, W).to(dev)
bg = torch.zeros(3, device=dev)
cams = [c.to(dev) for c in cameras.rotate360_path(W, H, n_views=30)]
rast = [GaussianRasterizer(GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), bg, 1.0,
        c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)) for c in cams]
config.set_async(True)
config.set_fused_grad_accumulation(True)
N = 20


def timed(name, fn, views):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:64s} host {1e6 * (t1 - t0) / (N * views):7.1f} us/view   wall {1e6 * (t2 - t0) / (N * views):7.1f} us/view", flush=True)


call = lambda r: r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                   rotations=leaf["rotations"])[0]
with config.overflow_policy("drop"):
    def fwd_nograd():
        with torch.no_grad():
            for r in rast:
                call(r)
    timed("operator forward, no_grad", fwd_nograd, 30)

    def fwd_grad():
        for r in rast:
            call(r)
    timed("operator forward with autograd node", fwd_grad, 30)

    def fwd_direct():
        for r in rast:
            parallel._direct_backward(call(r), g)
    timed("forward + backward node called directly, one stream", fwd_direct, 30)
for policy in ("recover", "drop"):
    pipe = parallel.ViewStreams(dev, 3, direct=True, on_overflow=policy)

    def step():
        pipe.begin_step()
        for r in rast:
            pipe.run_view(lambda r=r: call(r), grad_output=g)
        pipe.end_step()
    timed(f"ViewStreams(3, {policy}) step of 30 views, direct backward", step, 30)
batch = parallel.ViewBatch(cams, [g] * 30, 3, bg, 100_000, n_streams=3)
acc = {"means3D": leaf["means3D"].grad, "opacity": leaf["opacities"].grad, "scales": leaf["scales"].grad,
       "rotations": leaf["rotations"].grad, "sh": leaf["shs"].grad, "means2D": m2d.grad}
timed("lr_views_accumulate, 3 streams",
      lambda: batch.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(), leaf["rotations"].detach(),
                        leaf["shs"].detach(), acc), 30)
