"""Seeded synthetic Gaussian clouds (SURVEY.md section 8d "Synthetic inputs").

All tensors are generated on the CPU with torch.Generator().manual_seed(seed) so the oracle
and the HIP path see identical bits; callers move them to the device.
"""
import math
from typing import Dict

import torch


def make_cloud(P: int, kind: str = "box", seed: int = 0, sh_coeffs: int = 16,
               scale_mult: float = 1.0) -> Dict[str, torch.Tensor]:
    """Returns activated Gaussian attributes as the rasterizer consumes them:
    means3D (P,3), scales (P,3) > 0, rotations (P,4) unit quaternions (r,x,y,z),
    opacities (P,1) in (0,1), shs (P,sh_coeffs,3).

    kind == "box":  xyz = U(-1,1)^3 * (2.0, 1.2, 1.5) + (0,0,4)   (single view down +z)
    kind == "band": azimuth U(0,2pi), elevation U(-0.35,0.35), radius U(2,6) (camera paths
                    that rotate about the origin see a similar load from every direction)
    kind == "shell": one layer of pixel-sized Gaussians on a band around the origin (LucidDreamer's own scene statistics)
    """
    g = torch.Generator().manual_seed(seed)
    if kind == "box":
        u = torch.rand(P, 3, generator=g) * 2.0 - 1.0
        means = u * torch.tensor([2.0, 1.2, 1.5]) + torch.tensor([0.0, 0.0, 4.0])
    elif kind == "band":
        az = torch.rand(P, generator=g) * (2.0 * math.pi)
        el = (torch.rand(P, generator=g) * 2.0 - 1.0) * 0.35
        rad = 2.0 + 4.0 * torch.rand(P, generator=g)
        means = torch.stack([rad * torch.cos(el) * torch.sin(az),
                             rad * torch.sin(el),
                             rad * torch.cos(el) * torch.cos(az)], dim=1)
    elif kind == "shell":
        # The statistics of LucidDreamer's own scenes (R/luciddreamer.py: a point cloud lifted from RGB-D views of a panorama,
        # R/scene/gaussian_model.py:126-147 create_from_pcd: one isotropic Gaussian per point, its scale the distance to its
        # neighbours): ONE layer of points on a band around the camera, as dense as the pixels of a 512 x 512 view (a view of
        # the rotate360 path sees a quarter of them, about one per pixel), each about a pixel wide.
        az = torch.rand(P, generator=g) * (2.0 * math.pi)
        zz = (torch.rand(P, generator=g) * 2.0 - 1.0) * 0.3333                  # uniform in area on the band |sin(elevation)| < 1/3
        rad = 3.0 + 0.8 * torch.sin(3.0 * az) * torch.cos(5.0 * zz) + 0.05 * torch.randn(P, generator=g)
        c = torch.sqrt(1.0 - zz * zz)
        means = torch.stack([rad * c * torch.sin(az), rad * zz, rad * c * torch.cos(az)], dim=1)
        spacing = rad * math.sqrt(4.19 / float(max(P, 1)))                       # neighbour distance: band area 4.19 sr
        scales = (spacing * scale_mult).unsqueeze(1) * torch.exp(0.15 * torch.randn(P, 3, generator=g))
    else:
        raise ValueError(f"unknown cloud kind {kind!r}")
    s0 = 0.5 * float(max(P, 1)) ** (-1.0 / 3.0) * scale_mult
    if kind != "shell":
        scales = torch.exp(math.log(s0) + 0.3 * torch.randn(P, 3, generator=g))
    q = torch.randn(P, 4, generator=g)
    rotations = q / q.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g))
    shs = torch.empty(P, sh_coeffs, 3)
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) - 0.5) / 0.28209479177387814
    if sh_coeffs > 1:
        shs[:, 1:, :] = 0.1 * torch.randn(P, sh_coeffs - 1, 3, generator=g)
    return dict(means3D=means.float().contiguous(), scales=scales.float().contiguous(),
                rotations=rotations.float().contiguous(), opacities=opacities.float().contiguous(),
                shs=shs.float().contiguous())


def upstream_grad(height: int, width: int, seed: int = 1) -> torch.Tensor:
    """dL/dcolor = N(0,1) of shape (3,H,W)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, height, width, generator=g)
