"""Python operator of the MI355X rasterizer: same public API as the reference's
`depth_diff_gaussian_rasterization_min` package (RAST/depth_diff_gaussian_rasterization_min/__init__.py):

    GaussianRasterizationSettings   12-field NamedTuple, same names and order   (:158-170)
    GaussianRasterizer              nn.Module; forward(...) -> (color, radii, depth); markVisible (:172-221)
    rasterize_gaussians             functional entry                                (:21-42)

Error behaviour mirrored: plain `Exception` for bad SH/colour or scale/rotation/covariance
combinations (:192-196); with settings.debug the inputs of a failing call are dumped to
snapshot_fw.dump / snapshot_bw.dump before re-raising (:83-90, :133-140).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from . import config


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _take_ticket():
    """The early-header request of a forward is ALWAYS withdrawn, also when the forward raised: a request left
    behind would make an unrelated later forward on this thread post a ticket nobody takes.  If the forward failed after the
    ticket was posted, a ticket of the copy-and-event kind is waited for once, which releases it (log tickets hold nothing)."""
    import sys
    ticket = _C.take_early_ticket()
    if sys.exc_info()[0] is not None and 0 <= ticket < (1 << 40):
        try:
            _C.header_poll(ticket, True)
        except Exception:
            pass
        return -1
    return ticket


def _snapshot(args, path):
    torch.save(tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args), path)


class _RasterizeGaussians(torch.autograd.Function):
    """forward/backward wiring of RAST/.../__init__.py:44-156."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        capacity = config.capacity_for(means3D, rs)
        verifying = config.verifying(capacity)
        try:
            ticket = -1
            if verifying:
                _C.request_early_header()
            try:
                num_rendered, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
                    *args, binning_capacity=capacity)
            finally:
                if verifying:
                    ticket = _take_ticket()
            # policy "verify": every kernel of the forward is enqueued; wait for the copy of the header the library posted
            # after the scan, and if the view needs more instances than the buffer holds render it again in exact mode --
            # what is returned is always a complete image
            if verifying and config.verify(means3D, rs, ticket):
                capacity = 0
                num_rendered, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(*args, binning_capacity=0)
        except Exception:
            if rs.debug:
                _snapshot(args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise
        config.note_forward(means3D, rs, num_rendered, geom, capacity)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.binning_capacity = capacity
        # leaf inputs whose .grad already exists can be accumulated into in place by the backward kernel
        ctx.leaf_inputs = dict(means3D=means3D, means2D=means2D, sh=sh, colors=colors_precomp, opacity=opacities,
                               scales=scales, rotations=rotations, cov3D=cov3Ds_precomp) \
            if config.fused_grad_accumulation() else None
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, sh, rs.sh_degree,
                rs.campos, geom, ctx.num_rendered, binning, img, rs.debug)
        accumulate_into = None
        if ctx.leaf_inputs is not None:
            accumulate_into = {}
            for name, t in ctx.leaf_inputs.items():
                g = t.grad if (t.is_leaf and t.requires_grad and t.numel() != 0) else None
                if g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == means3D.device \
                        and g.data_ptr() % 16 == 0:      # the kernels accumulate with 16-byte accesses; else: dense path
                    accumulate_into[name] = g
        try:
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(
                *args, binning_capacity=ctx.binning_capacity, accumulate_into=accumulate_into, skip_unused=True)
        except Exception:
            if rs.debug:
                _snapshot(args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
            raise
        # order of the forward inputs (RAST/.../__init__.py:144-154); gradients of absent inputs are None-able
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """The operator.  Its autograd node is compiled (csrc/torch_ext.cpp RasterizeFn: forward and backward run without the
    interpreter -- the Python node below cost more host time per 1080p view than the GPU needs for it); with settings.debug
    the Python node runs instead, because it is the one that writes the reference's snapshot_fw.dump / snapshot_bw.dump."""
    rs = raster_settings
    if rs.debug:
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, rs)
    capacity = config.capacity_for(means3D, rs)
    verifying = config.verifying(capacity)
    fused = config.fused_grad_accumulation()
    offered = config.offered_grad_output()
    if offered is not None and fused and not verifying:
        # the caller already holds dL/dcolor (parallel.ViewStreams.run_view): forward and backward in ONE call of the binding,
        # gradients added into the leaves' .grad by the kernels, no autograd node (csrc/torch_ext.cpp rasterize_view_step).
        # An input that is not a leaf with a suitable .grad -> empty result -> the autograd path below, offer untouched.
        out = _C.rasterize_view_step(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg,
                                     rs.viewmatrix, rs.projmatrix, rs.campos, rs.scale_modifier, rs.tanfovx, rs.tanfovy,
                                     rs.image_height, rs.image_width, rs.sh_degree, rs.prefiltered, capacity, offered)
        if out:
            color, radii, depth, geom = out
            config.mark_grad_output_taken()
            config.note_forward(means3D, rs, _C.last_num_rendered(), geom, capacity)
            return color, radii, depth

    def run(cap):
        return _C.rasterize_autograd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg,
                                     rs.viewmatrix, rs.projmatrix, rs.campos, rs.scale_modifier, rs.tanfovx, rs.tanfovy,
                                     rs.image_height, rs.image_width, rs.sh_degree, rs.prefiltered, cap, fused)
    ticket = -1
    if verifying:
        _C.request_early_header()
    try:
        color, radii, depth, geom = run(capacity)
    finally:
        if verifying:
            ticket = _take_ticket()
    # policy "verify" (config.py): the whole forward is enqueued; if the view needs more instances than its buffer holds it is
    # rendered again in exact mode -- what is returned is always a complete image (the first node is simply dropped)
    if verifying and config.verify(means3D, rs, ticket):
        capacity = 0
        color, radii, depth, geom = run(0)
    config.note_forward(means3D, rs, _C.last_num_rendered(), geom, capacity)
    return color, radii, depth


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Autograd op over the STORED GaussianModel tensors: one forward and one backward call replace
    exp / normalize / sigmoid / cat and their autograd nodes (R/scene/gaussian_model.py:97-117)."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings):
        rs = raster_settings
        capacity = config.capacity_for(xyz, rs)
        verifying = config.verifying(capacity)

        def run(cap):
            return _C.rasterize_gaussians_raw(
                rs.bg, xyz, features_dc, features_rest, opacity, scaling, rotation, rs.scale_modifier, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug,
                binning_capacity=cap)
        ticket = -1
        if verifying:
            _C.request_early_header()
        try:
            num_rendered, color, depth, radii, geom, binning, img = run(capacity)
        finally:
            if verifying:
                ticket = _take_ticket()
        if verifying and config.verify(xyz, rs, ticket):       # overflowed: an exact-mode render instead
            capacity = 0
            num_rendered, color, depth, radii, geom, binning, img = run(0)
        config.note_forward(xyz, rs, num_rendered, geom, capacity)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.binning_capacity = capacity
        ctx.leaf_inputs = dict(xyz=xyz, means2D=means2D, features_dc=features_dc, features_rest=features_rest,
                               opacity=opacity, scaling=scaling, rotation=rotation) \
            if config.fused_grad_accumulation() else None
        ctx.save_for_backward(xyz, features_dc, features_rest, opacity, scaling, rotation, radii, geom, binning, img)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        xyz, features_dc, features_rest, opacity, scaling, rotation, radii, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=xyz.device)
        accumulate_into = None
        if ctx.leaf_inputs is not None:
            def leaf_grad(t):
                g = t.grad if (t.is_leaf and t.requires_grad and t.numel() != 0) else None
                ok = g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == xyz.device \
                    and g.data_ptr() % 16 == 0
                return g if ok else None
            li = ctx.leaf_inputs
            accumulate_into = {k: leaf_grad(li[k]) for k in ("xyz", "means2D", "opacity", "scaling", "rotation")}
            g_dc, g_rest = leaf_grad(li["features_dc"]), leaf_grad(li["features_rest"])
            if g_dc is not None and (g_rest is not None or features_rest.numel() == 0):
                accumulate_into["features"] = (g_dc, g_rest)
        # an armed FusedAdam over exactly these six tensors (optim.FusedAdam.arm_fused_backward): the gradients go to tensors
        # autograd never sees -- visited rows only, nothing zero-filled -- and the optimizer's masked step
        # (lr_adam_step_masked: the view's own visibility is the mask) is launched right behind the backward's kernels;
        # optimizer.step() later only checks the iteration.  param.grad stays None.
        from . import optim
        opt = optim.take_armed((xyz, features_dc, features_rest, opacity, scaling, rotation)) \
            if (optim._armed is not None and not rs.debug and all(ctx.needs_input_grad[k] for k in (0, 2, 4, 5, 6))) else None
        if opt is not None:
            g = _C.rasterize_gaussians_raw_backward(
                rs.bg, xyz, radii, features_dc, features_rest, opacity, scaling, rotation, rs.scale_modifier, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, rs.sh_degree, rs.campos, geom, ctx.num_rendered,
                binning, img, False, binning_capacity=ctx.binning_capacity, no_zero_fill=True)
            opt.apply_armed_step(geom, [xyz, features_dc, features_rest, opacity, scaling, rotation], list(g[1:]))
            return None, g[0], None, None, None, None, None, None
        g = _C.rasterize_gaussians_raw_backward(
            rs.bg, xyz, radii, features_dc, features_rest, opacity, scaling, rotation, rs.scale_modifier, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, rs.sh_degree, rs.campos, geom, ctx.num_rendered,
            binning, img, rs.debug, binning_capacity=ctx.binning_capacity, accumulate_into=accumulate_into)
        g_means2D, g_xyz, g_dc, g_rest, g_op, g_sc, g_rot = g
        return g_xyz, g_means2D, g_dc, g_rest, g_op, g_sc, g_rot, None


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings):
    """(color, radii, depth) from the stored GaussianModel parameters (pre-activation), see _RasterizeGaussiansRaw."""
    return _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                                        raster_settings)


_EMPTY = torch.Tensor([])           # never written: one instance serves every call (three constructions per call were ~6 us)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points in front of the near plane (view z > 0.2)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        scale_rot_given = scales is not None or rotations is not None
        scale_rot_complete = scales is not None and rotations is not None
        if (not scale_rot_complete and cov3D_precomp is None) or (scale_rot_given and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        empty = _EMPTY                  # the reference's placeholder for an absent optional input (`torch.Tensor([])`, :198-208)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
