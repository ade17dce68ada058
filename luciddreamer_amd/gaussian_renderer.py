"""render(viewpoint_camera, pc, opt, bg_color, ...) -- same call contract as the reference's
/root/reference/gaussian_renderer/__init__.py:18-104, on top of the MI355X rasterizer.

LucidDreamer itself keeps using ITS OWN gaussian_renderer module unchanged (it only needs the
`depth_diff_gaussian_rasterization_min` import to resolve to this repository).  This copy exists so
that the bench, the data-parallel wrapper and the tests can drive the hot path the way the training
loop does (/root/reference/luciddreamer.py:251, 296) without importing the reference.  `pc` is
duck-typed: anything with the getters of scene/gaussian_model.py:97-120 works (e.g. GaussianCloud below).
"""
import math
from types import SimpleNamespace

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw
from .sh import colors_from_shs

DEFAULT_OPT = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)


def render(viewpoint_camera, pc, opt=DEFAULT_OPT, bg_color=None, scaling_modifier=1.0, override_color=None,
           render_only=False):
    """Returns {"render", "viewspace_points", "visibility_filter", "radii", "depth"} (or render/depth only)."""
    xyz = pc.get_xyz
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=xyz.device)
    # zero tensor whose gradient is the screen-space (NDC-scaled) mean gradient used by densification
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=bool(getattr(opt, "debug", False)),
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = None
    if getattr(opt, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(opt, "convert_SHs_python", False):
            colors_precomp = colors_from_shs(pc, viewpoint_camera.camera_center)      # :73-78
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, depth = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
        opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    if render_only:
        return {"render": rendered_image, "depth": depth}
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth}


def render_raw(viewpoint_camera, pc, opt=DEFAULT_OPT, bg_color=None, scaling_modifier=1.0, render_only=False):
    """render() for the common training configuration (SH colours, scale/rotation covariance), reading the STORED
    parameters of `pc` (`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`,
    /root/reference/scene/gaussian_model.py:47-52) instead of the activated getters: the activations and the
    dc|rest concatenation happen inside the rasterizer kernels.  Same return dict as render()."""
    xyz = pc._xyz
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=xyz.device)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=bool(getattr(opt, "debug", False)))
    rendered_image, radii, depth = rasterize_gaussians_raw(xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                           pc._opacity, pc._scaling, pc._rotation, rs)
    if render_only:
        return {"render": rendered_image, "depth": depth}
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth}


class GaussianCloud:
    """Minimal parameter holder with GaussianModel's getters (scene/gaussian_model.py:97-120):
    raw parameters are log-scales, logit-opacities, unnormalised quaternions, SH split in dc/rest."""

    def __init__(self, means3D, scales, rotations, opacities, shs, active_sh_degree=3, requires_grad=True):
        eps = 1e-6
        self.max_sh_degree = int(round(math.sqrt(shs.shape[1]))) - 1
        self.active_sh_degree = active_sh_degree
        mk = lambda t: t.detach().clone().contiguous().requires_grad_(requires_grad)
        self._xyz = mk(means3D)
        self._features_dc = mk(shs[:, :1, :])
        self._features_rest = mk(shs[:, 1:, :])
        self._scaling = mk(torch.log(scales))
        self._rotation = mk(rotations)
        op = opacities.clamp(eps, 1 - eps)
        self._opacity = mk(torch.log(op / (1 - op)))

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)
