"""The per-iteration body of LucidDreamer's training loop (R/luciddreamer.py:283-327) around the reference's OWN
classes, imported unchanged through oracle/ref_python.py: GaussianModel (parameters, Adam groups, learning-rate schedule,
densification statistics, densify_and_prune), render() and utils/loss.py's l1_loss / ssim.  The same function drives

  * the device run: rasterizer = this repository's packages (the drop-in under test), tensors on cuda:0;
  * the checker run: rasterizer = a CPU backend of the oracle ("port": the restatement, "ref": oracle/_ref), tensors on
    the CPU.

Test infrastructure.  Only the loop skeleton is written here (camera choice is a seeded sequence instead of randint;
targets are given); everything it calls is the reference's code."""
import contextlib

import numpy as np
import torch

from oracle import ref_python as rp


def model_from_cloud(R, cloud, device, sh_degree=3, active_sh_degree=None):
    """A reference GaussianModel whose stored (pre-activation) parameters reproduce `cloud` (activated attributes)."""
    gm = R.gaussian_model.GaussianModel(sh_degree)
    mk = lambda t: torch.nn.Parameter(t.to(device).contiguous().requires_grad_(True))
    gm._xyz = mk(cloud["means3D"].clone())
    gm._features_dc = mk(cloud["shs"][:, 0:1, :].clone())
    gm._features_rest = mk(cloud["shs"][:, 1:, :].clone())
    gm._scaling = mk(torch.log(cloud["scales"]))
    gm._rotation = mk(cloud["rotations"].clone())
    gm._opacity = mk(torch.logit(cloud["opacities"].clamp(1e-4, 1 - 1e-4)))
    gm.max_radii2D = torch.zeros(cloud["means3D"].shape[0], device=device)
    gm.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
    gm.spatial_lr_scale = 1.0
    return gm


def train(R, gm, device, cams, order, targets, depth_targets=None, iters=200, depth_weight=0.1, densify_from=10 ** 9,
          densify_every=100, extent=3.0, opt=None, on_iteration=None, on_loss=None, seed=0):
    """luciddreamer.py:283-327 with `order[it]` as the camera index.  Returns dict(loss=[...], P=[...])."""
    opt = opt or R.arguments.GSParams()
    opt.iterations = iters + 1                      # the reference skips the optimizer step at the last iteration (:325)
    # the reference reads its densification schedule from `opt` (:307-317); the harness's arguments go there too, so that what
    # GaussianModel.training_setup(opt) sees IS the schedule the loop below runs (install(fuse_step=True) arms by it)
    opt.densify_from_iter, opt.densification_interval = densify_from, densify_every
    render, l1_loss, ssim = R.gaussian_renderer.render, R.loss.l1_loss, R.loss.ssim
    bg = torch.zeros(3, device=device)              # R/arguments.py:14 white_background False
    if gm.optimizer is None:
        gm.training_setup(opt)
    cams_d = [c.to(device) for c in cams]
    tg = [t.to(device) for t in targets]
    dg = None if depth_targets is None else [t.to(device) for t in depth_targets]
    losses, counts = [], []
    torch.manual_seed(seed)
    for iteration in range(1, iters + 1):
        gm.update_learning_rate(iteration)                                         # :284
        if iteration % 1000 == 0:
            gm.oneupSHdegree()
        k = order[iteration - 1]
        pkg = render(cams_d[k], gm, opt, bg)                                       # :296
        image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
        Ll1 = l1_loss(image, tg[k])                                                # :301-303
        if opt.lambda_dssim == 0.0:             # L1 only (tests that need run-to-run bit-repeatability: MIOpen may pick
            loss = Ll1                          # a different convolution algorithm for ssim's grouped conv on a later call)
        else:
            loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, tg[k]))
        if dg is not None:       # a depth term: enters the loss, contributes no parameter gradient (backward.cu:539-554)
            loss = loss + depth_weight * l1_loss(pkg["depth"], dg[k])
        if on_loss is not None:
            on_loss(iteration, gm, pkg, loss, k)
        loss.backward()                                                            # :304
        with torch.no_grad():
            gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])       # :310-312
            gm.add_densification_stats(vsp, vis)
            if iteration < opt.densify_until_iter and iteration > opt.densify_from_iter \
                    and iteration % opt.densification_interval == 0:               # :307, :314-317
                gm.densify_and_prune(opt.densify_grad_threshold, 0.005, extent, None)
            if iteration < opt.iterations:                                         # :325-327
                gm.optimizer.step()
                gm.optimizer.zero_grad(set_to_none=True)
        # the reference's loop does not read the loss (R/luciddreamer.py:283-327 has no .item()): the values are kept on the
        # device and fetched after the last iteration, so that the harness adds no host <-> device round trip of its own
        losses.append(loss.detach())
        counts.append(int(gm.get_xyz.shape[0]))
        if on_iteration is not None:
            on_iteration(iteration, gm, pkg, loss)
    return dict(loss=np.array([float(v) for v in losses]), P=np.array(counts))


def resident(R, gm, device, cams, targets, depth_targets=None, iters=200, opt=None):
    """What the reference has in place BEFORE its loop starts, so that a timed train() is iterations only: cameras and target
    images on the device (R/scene/cameras.py keeps original_image on data_device = cuda) and the optimizer built
    (GaussianModel.training_setup is called once, before the loop: R/luciddreamer.py:279).  Returns (cams, targets, depths, opt)
    to pass to train()."""
    opt = opt or R.arguments.GSParams()
    opt.iterations = iters + 1
    opt.densify_from_iter = 10 ** 9                 # (train()'s default: no densification unless the caller asks for it)
    if gm.optimizer is None:
        gm.training_setup(opt)
    out = ([c.to(device) for c in cams], [t.to(device) for t in targets],
           None if depth_targets is None else [t.to(device) for t in depth_targets], opt)
    # everything the set-up created (modules, model, targets: ~10^5 Python objects) is still in the collector's young
    # generations; the first collection inside a short timed loop would walk all of it -- 50-90 ms, i.e. +0.6-1.1 ms per
    # iteration of an 80-iteration pass, in some passes and not in others (profiles/r04q_loop_drift.txt).  A training run pays
    # that once; a timed pass must not: collect now, so that what survives is old
    import gc
    gc.collect()
    return out


@contextlib.contextmanager
def stack(rasterizer):
    """reference modules + (for CPU backends) the cuda->cpu mapping, as one context."""
    with rp.reference_modules(rasterizer) as R:
        if rasterizer in ("ours", "refdev"):
            yield R, torch.device("cuda:0")
        else:
            with rp.cuda_as_cpu():
                yield R, torch.device("cpu")
