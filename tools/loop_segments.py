#!/usr/bin/env python
"""Host time per segment of the UNCHANGED reference iteration after luciddreamer_amd.install() (1 M Gaussians, 512 x 512): the
body of tests/ref_loop.train with a clock between the statements -- once free-running (host time to issue) and once with a
device synchronize after every segment (host + GPU time of the segment)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import luciddreamer_amd                                  # noqa: E402
from luciddreamer_amd import cameras, config             # noqa: E402
from tests import ref_loop                                # noqa: E402
from tests.test_gpu_reference_stack import _perturbed, _targets   # noqa: E402

P, W, H, iters = 1_000_000, 512, 512, 60
cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
base, hidden = _perturbed(P, 41)
targets, depths = _targets(hidden, cams)
order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=iters)]
config.reset()
config.set_async(True, on_overflow=os.environ.get("LR_POLICY", "verify"))
if os.environ.get("LR_SINGLE_THREAD_BACKWARD"):
    torch.autograd.set_multithreading_enabled(False)      # backward nodes on the calling thread: no hand-off to the engine's thread
if "--hand" in sys.argv:
    # the hand-edited iteration of bench.py's `with_optional_pieces` leg, same clock
    import importlib.util
    spec = importlib.util.spec_from_file_location("lr_example_train_loop", os.path.join(ROOT, "examples", "train_loop.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    from luciddreamer_amd import densify
    from luciddreamer_amd.gaussian_renderer import render_raw
    from luciddreamer_amd.loss import l1_dssim_loss
    dev = torch.device("cuda:0")
    cams_d = [c.to(dev) for c in cams]
    tg, dg = [t.to(dev) for t in targets], [t.to(dev) for t in depths]
    bg = torch.zeros(3, device=dev)
    for sync in (False, False, True):
        b = {k: v.to(dev) for k, v in base.items()}
        model = ex.TrainableCloud(b["means3D"], b["scales"], b["rotations"], b["opacities"], b["shs"])
        model.training_setup({"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3})
        seg = {}
        torch.cuda.synchronize()
        t_all = time.perf_counter()

        def mark(name, t0):
            if sync:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            seg[name] = seg.get(name, 0.0) + (t1 - t0)
            return t1
        for it in range(iters):
            t = time.perf_counter()
            k = order[it]
            pkg = render_raw(cams_d[k], model, bg_color=bg)
            t = mark("render", t)
            loss = l1_dssim_loss(pkg["render"], tg[k], 0.2) + 0.1 * (pkg["depth"] - dg[k]).abs().mean()
            t = mark("loss", t)
            loss.backward()
            t = mark("backward", t)
            with torch.no_grad():
                densify.add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
                t = mark("add_densification_stats", t)
                model.optimizer.step()
                t = mark("optimizer.step", t)
                model.optimizer.zero_grad(set_to_none=True)
                t = mark("zero_grad", t)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t_all) / iters * 1e3
        print(("with a synchronize after every segment" if sync else "free running") + f": {total:.3f} ms / iteration; segments (ms): " +
              ", ".join(f"{k} {v / iters * 1e3:.3f}" for k, v in seg.items()), flush=True)
    sys.exit(0)
# host time inside the two custom backward functions (they run on the autograd engine's thread: the main thread only sees the wait)
inside = {}


def clocked(cls, label):
    orig = cls.backward

    def backward(ctx, *a):
        t0 = time.perf_counter()
        try:
            return orig(ctx, *a)
        finally:
            inside[label] = inside.get(label, 0.0) + time.perf_counter() - t0
    cls.backward = staticmethod(backward)


from luciddreamer_amd import loss as _loss_mod, rasterizer as _rast_mod   # noqa: E402
clocked(_rast_mod._RasterizeGaussiansRaw, "rasterizer backward")
clocked(_loss_mod._L1SSIMPair, "loss pair backward")
with ref_loop.stack("ours") as (R, dev):
    h = luciddreamer_amd.install(R) if "--plain" not in sys.argv else None
    try:
        for sync in (False, False, True):
            gm = ref_loop.model_from_cloud(R, base, dev)
            cams_d, tg, dg, opt = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
            render, l1_loss, ssim = R.gaussian_renderer.render, R.loss.l1_loss, R.loss.ssim
            bg = torch.zeros(3, device=dev)
            seg = {}
            inside.clear()
            torch.cuda.synchronize()
            t_all = time.perf_counter()

            def mark(name, t0):
                if sync:
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                seg[name] = seg.get(name, 0.0) + (t1 - t0)
                return t1
            for it in range(1, iters + 1):
                t = time.perf_counter()
                gm.update_learning_rate(it)
                t = mark("update_learning_rate", t)
                k = order[it - 1]
                pkg = render(cams_d[k], gm, opt, bg)
                image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
                t = mark("render", t)
                Ll1 = l1_loss(image, tg[k])
                loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, tg[k]))
                loss = loss + 0.1 * l1_loss(pkg["depth"], dg[k])
                t = mark("loss", t)
                loss.backward()
                t = mark("backward", t)
                with torch.no_grad():
                    gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])
                    t = mark("max_radii line", t)
                    gm.add_densification_stats(vsp, vis)
                    t = mark("add_densification_stats", t)
                    gm.optimizer.step()
                    t = mark("optimizer.step", t)
                    gm.optimizer.zero_grad(set_to_none=True)
                    t = mark("zero_grad", t)
            torch.cuda.synchronize()
            total = (time.perf_counter() - t_all) / iters * 1e3
            print(("with a synchronize after every segment" if sync else "free running") + f": {total:.3f} ms / iteration; segments (ms): " +
                  ", ".join(f"{k} {v / iters * 1e3:.3f}" for k, v in seg.items()) +
                  "; inside: " + ", ".join(f"{k} {v / iters * 1e3:.3f}" for k, v in inside.items()), flush=True)
    finally:
        if h is not None:
            luciddreamer_amd.uninstall(h)
