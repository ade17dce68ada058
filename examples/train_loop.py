#!/usr/bin/env python
"""A LucidDreamer-style optimisation loop (/root/reference/luciddreamer.py:221-327) on synthetic data, using every
piece of this repository on the MI355X:

    render_raw        rasterizer fed with the stored parameters (activations inside the kernels)      8a-8d, 8f-2
    l1_dssim_loss     fused (1-l)*L1 + l*(1-SSIM) and its gradient                                     8f-3
    densify           densify_and_prune / prune through one row-selection kernel, Adam state intact     8f-4
    FusedAdam         torch.optim.Adam's arithmetic in one launch per step                              (8e: the step after the all-reduce)
    distCUDA2         initial scales from the 3-nearest-neighbour distance                              8f-1

Targets are renders of a hidden "ground truth" cloud from a look-around camera path; the trained cloud starts from
a perturbed subset of it.  Prints the loss every `--log` iterations and the time per iteration.

    python examples/train_loop.py [--gaussians 200000] [--iters 300] [--resolution 512x512]
"""
import argparse
import math
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luciddreamer_amd import cameras, config, densify, synthetic           # noqa: E402
from luciddreamer_amd.gaussian_renderer import GaussianCloud, render_raw   # noqa: E402
from luciddreamer_amd.loss import l1_dssim_loss                            # noqa: E402
from luciddreamer_amd.optim import FusedAdam                               # noqa: E402
from simple_knn._C import distCUDA2                                        # noqa: E402

GROUP_ATTR = densify.GROUP_ATTR


class TrainableCloud(GaussianCloud):
    """GaussianCloud + what GaussianModel.training_setup adds (scene/gaussian_model.py:148-169)."""

    def training_setup(self, lrs, percent_dense=0.01, torch_adam=False):
        P = self._xyz.shape[0]
        dev = self._xyz.device
        for a in GROUP_ATTR.values():
            setattr(self, a, torch.nn.Parameter(getattr(self, a).detach().clone().requires_grad_(True)))
        self.percent_dense = percent_dense
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)
        groups = [{"params": [getattr(self, a)], "lr": lrs[n], "name": n} for n, a in GROUP_ATTR.items()]
        self.optimizer = (torch.optim.Adam if torch_adam else FusedAdam)(groups, lr=0.0, eps=1e-15)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):     # gaussian_model.py:405-407
        # same sums as the reference's boolean-mask indexing, written without the host synchronisation that indexing
        # with a mask implies (nonzero): rows outside the filter add 0
        f = update_filter[:, None].float()
        self.xyz_gradient_accum += f * torch.norm(viewspace_point_tensor.grad[:, :2], dim=-1, keepdim=True)
        self.denom += f


def build(args, dev):
    W, H = (int(v) for v in args.resolution.split("x"))
    gt_cloud = {k: v.to(dev) for k, v in synthetic.make_cloud(args.gaussians, "box", 0).items()}
    gt = GaussianCloud(gt_cloud["means3D"], gt_cloud["scales"], gt_cloud["rotations"], gt_cloud["opacities"], gt_cloud["shs"],
                       requires_grad=False)
    cams = [c.to(dev) for c in cameras.lookaround_path(W, H, n_views=args.views)]
    with torch.no_grad():
        targets = [render_raw(c, gt, render_only=True)["render"].clamp(0, 1) for c in cams]
    # start: every second Gaussian, jittered; scales from the 3-NN distance like create_from_pcd (gaussian_model.py:136-137)
    g = torch.Generator(device="cpu").manual_seed(1)
    idx = torch.arange(0, args.gaussians, 2, device=dev)
    xyz = gt_cloud["means3D"][idx] + 0.01 * torch.randn(idx.numel(), 3, generator=g).to(dev)
    dist2 = torch.clamp_min(distCUDA2(xyz.contiguous()), 1e-7)
    scales = torch.sqrt(dist2)[:, None].repeat(1, 3)
    rots = torch.zeros(idx.numel(), 4, device=dev)
    rots[:, 0] = 1
    shs = torch.zeros(idx.numel(), 16, 3, device=dev)
    shs[:, 0] = gt_cloud["shs"][idx, 0]
    model = TrainableCloud(xyz, scales, rots, torch.full((idx.numel(), 1), 0.1, device=dev), shs)
    model.training_setup({"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3},
                         torch_adam=args.torch_adam)
    return model, cams, targets


def train(args, log=print):
    dev = torch.device("cuda:0")
    model, cams, targets = build(args, dev)
    bg = torch.zeros(3, device=dev)
    # no host round trip per forward ("Host sync"); instance counts drift while the cloud is
    # optimised, so the capacity is taken over the views of the path, with generous headroom
    config.set_async(not args.exact, headroom=1.5, warm_calls=len(cams))     # overflowed views are re-rendered (default policy)
    losses = []
    gen = torch.Generator().manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, args.iters + 1):
        v = int(torch.randint(0, len(cams), (1,), generator=gen))
        pkg = render_raw(cams[v], model, bg_color=bg)                                     # luciddreamer.py:296
        loss = l1_dssim_loss(pkg["render"], targets[v], args.lambda_dssim)                # :301-303
        loss.backward()                                                                   # :304
        with torch.no_grad():
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            if it < args.densify_until:                                                   # :308-318
                densify.add_densification_stats(model, pkg["viewspace_points"], radii)   # :310-311 + stats, one kernel
                if it >= args.densify_from and it % args.densify_every == 0:
                    densify.densify_and_prune(model, 0.0002, 0.005, 5.0, 20)
            model.optimizer.step()                                                        # :322-324
            model.optimizer.zero_grad(set_to_none=True)
        if it % args.log == 0 or it == 1:
            losses.append((it, float(loss.item()), int(model._xyz.shape[0])))
            log(f"iter {it:5d}  loss {losses[-1][1]:.5f}  gaussians {losses[-1][2]}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    config.set_async(True)              # back to the library defaults
    log(f"{args.iters} iterations in {dt:.2f} s = {dt / args.iters * 1e3:.3f} ms/iteration")
    return losses, dt


def default_args(**kw):
    d = dict(gaussians=200_000, iters=300, resolution="512x512", views=12, lambda_dssim=0.2, log=50, densify_from=100,
             densify_every=100, densify_until=10_000, exact=False, torch_adam=False)
    d.update(kw)
    return SimpleNamespace(**d)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    for k, v in vars(default_args()).items():
        if isinstance(v, bool):
            ap.add_argument("--" + k.replace("_", "-"), action="store_true")
        else:
            ap.add_argument("--" + k.replace("_", "-"), type=type(v), default=v)
    train(ap.parse_args())
