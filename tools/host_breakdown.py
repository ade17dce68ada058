#!/usr/bin/env python
"""Where does the host time of the drop-in operator go?  C3, per view, host time to ENQUEUE (the GPU is drained only between
the legs): the library's forward / backward alone (C calls on preallocated buffers), the compiled binding without and with
an autograd node, forward + backward through autograd, and the same under parallel.ViewStreams."""
import ctypes
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import _lib, cameras, config, parallel, synthetic  # noqa: E402
from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = torch.device("cuda:0")
P, W, H = 1_000_000, 1920, 1080
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
N = 10


def timed(name, fn, views):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:58s} host {1e6 * (t1 - t0) / (N * views):7.1f} us/view   wall {1e6 * (t2 - t0) / (N * views):7.1f} us/view", flush=True)


call = lambda r: r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                   rotations=leaf["rotations"])[0]
with config.overflow_policy("drop"):
    def fwd_nograd():
        with torch.no_grad():
            for r in rast:
                call(r)
    timed("operator forward, no_grad (policy drop)", fwd_nograd, 30)

    def fwd_grad():
        for r in rast:
            call(r)
    timed("operator forward with autograd node (no backward)", fwd_grad, 30)

    def fwd_bwd():
        for r in rast:
            call(r).backward(g)
    timed("operator forward + backward, one stream", fwd_bwd, 30)
for ns, direct, policy in ((1, False, "recover"), (3, False, "recover"), (3, True, "recover"), (3, True, "drop")):
    pipe = parallel.ViewStreams(dev, ns, direct=direct, on_overflow=policy)

    def step():
        pipe.begin_step()
        for r in rast:
            pipe.run_view(lambda r=r: call(r), lambda c: c.backward(g))
        pipe.end_step()
    if not direct:
        timed(f"ViewStreams({ns}) step of 30 views, backward per view", step, 30)

    def step_grouped():
        pipe.begin_step()
        for r in rast:
            pipe.run_view(lambda r=r: call(r), grad_output=g)
        pipe.end_step()
    timed(f"ViewStreams({ns}, {policy}) 30 views, " + ("backward node called directly" if direct else "one engine pass per 6 views"), step_grouped, 30)
# the library alone: lr_views_accumulate (one C call per step)
batch = parallel.ViewBatch(cams, [g] * 30, 3, bg, 1_000_000, n_streams=3)
acc = {"means3D": leaf["means3D"].grad, "opacity": leaf["opacities"].grad, "scales": leaf["scales"].grad,
       "rotations": leaf["rotations"].grad, "sh": leaf["shs"].grad, "means2D": m2d.grad}
timed("lr_views_accumulate, 3 streams (the headline's entry point)",
      lambda: batch.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(), leaf["rotations"].detach(),
                        leaf["shs"].detach(), acc), 30)
