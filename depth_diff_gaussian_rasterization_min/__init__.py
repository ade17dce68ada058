"""Drop-in import name of the reference extension package
(RAST/depth_diff_gaussian_rasterization_min/__init__.py; imported by
/root/reference/gaussian_renderer/__init__.py:14).  Everything is implemented in luciddreamer_amd."""
from luciddreamer_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                         rasterize_gaussians, _RasterizeGaussians)
from luciddreamer_amd import _C  # noqa: F401
