"""Spherical-harmonics colour evaluation in torch, for render()'s `convert_SHs_python` branch
(/root/reference/gaussian_renderer/__init__.py:73-78, which calls /root/reference/utils/sh.py:57-112 eval_sh).

The rasterizer kernels evaluate the same real SH basis (bands 0..3, constants of RAST/cuda_rasterizer/auxiliary.h:22-39)
per Gaussian; this module exists for callers that ask for the Python route (colours then enter the rasterizer as
colors_precomp and the SH gradient flows through torch autograd).  Written as basis-matrix times coefficients:
one [P, K] basis tensor, one contraction -- no per-band Python arithmetic on [P, 3] tensors.
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """dirs [..., 3] unit vectors -> basis values [..., (deg+1)^2] in the reference's coefficient order."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, C0)]
    if deg > 0:
        cols += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        cols += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                 C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                 C3[6] * x * (xx - 3 * yy)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Same signature as the reference's eval_sh: sh [..., C, (max_deg+1)^2], dirs [..., 3] -> [..., C]."""
    k = (deg + 1) ** 2
    if sh.shape[-1] < k:
        raise ValueError("not enough SH coefficients for the requested degree")
    return (sh[..., :k] * sh_basis(deg, dirs).unsqueeze(-2)).sum(dim=-1)


def colors_from_shs(pc, camera_center: torch.Tensor) -> torch.Tensor:
    """gaussian_renderer/__init__.py:74-78: view-dependent RGB of every Gaussian, clamped at 0 after the +0.5 shift."""
    feats = pc.get_features                                              # [P, M, 3]
    shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
    d = pc.get_xyz - camera_center.reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, d) + 0.5, 0.0)
