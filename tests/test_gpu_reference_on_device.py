"""GPU: the HIP path against the REFERENCE'S OWN KERNELS running on the same MI355X -- oracle/_ref/libref_raster_gfx950.so,
forward.cu / backward.cu / rasterizer_impl.cu / simple_knn.cu compiled by hipcc from /root/reference in the build container
(oracle/build_ref.py build_device()).  Full problem sizes at device speed: C3 (1 M Gaussians, 1080p) on several views of
the path and C4's shape (3 M, 1440p).  The reference's per-pixel float atomics make its gradients run-to-run different, so
the gradient bar is the usual 1e-4 of each tensor's maximum, with at most 8 rows per tensor (measured: 0-2) that sit on a
discrete-threshold pixel up to 1.5e-3 (tests/test_gpu_full.py explains them); images 1e-5 outside at most 16 (C3, of 2.07 M;
measured 0-3) / 64 (C4 shape, of 3.7 M; measured 8) threshold pixels -- counts and the whole-image maximum are printed."""
import math

import numpy as np
import pytest
import torch

from luciddreamer_amd import cameras, synthetic
from oracle import ref_device
from tests import helpers as hp

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_device.available(), reason="oracle/_ref gfx950 build did not travel")]


def _both(cloud, cam, dev, g, degree=3):
    bg = torch.zeros(3)
    hip = hp.run_hip(cloud, cam, degree, bg, dev, g)
    c = {k: v.to(dev).contiguous() for k, v in cloud.items()}
    cd = cam.to(dev)
    r = ref_device.Renderer()
    tfx, tfy = hp.tan_fov(cam)
    R, color, depth, radii = r.forward(bg.to(dev), c["means3D"], None, c["opacities"], c["scales"], c["rotations"], 1.0, None,
                                       cd.world_view_transform.contiguous(), cd.full_proj_transform.contiguous(), tfx, tfy,
                                       cam.image_height, cam.image_width, c["shs"], degree, cd.camera_center.contiguous())
    grads = r.backward(g.to(dev).contiguous())
    return hip, R, color.cpu().numpy(), depth.cpu().numpy(), radii.cpu().numpy(), [t.cpu().numpy() for t in grads]


def _compare(hip, color, depth, radii, grads, P, max_bad_pixels, label):
    assert np.array_equal(hip["radii"], radii), label
    cerr = np.abs(hip["color"] - color).max(axis=0)
    derr = np.abs(hip["depth"][0] - depth[0]) / np.maximum(1.0, np.abs(depth[0]))
    bad = int(((cerr > hp.COLOR_ATOL) | (derr > hp.DEPTH_RTOL)).sum())
    print(label, f"pixels beyond 1e-5: {bad} of {cerr.size} (budget {max_bad_pixels}); whole-image max-abs colour error "
                 f"{cerr.max():.2e}, depth {derr.max():.2e}")
    assert bad <= max_bad_pixels, (label, bad)
    report = {}
    for k, b in zip(("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations"), grads):
        if k in ("colors", "cov3D"):
            continue
        a, b = hip["grads"][k].reshape(P, -1), b.reshape(P, -1)
        scale = float(np.abs(b).max())
        row = np.abs(a - b).max(axis=1)
        nbad = int((row > hp.GRAD_RTOL * scale).sum())
        report[k] = (f"{row.max() / scale:.1e}", nbad)
        assert nbad <= 8 and row.max() <= 1.5e-3 * scale, (label, k, nbad, row.max() / scale)
    print(label, "pixels beyond 1e-5:", bad, "gradient rows beyond 1e-4:", report)


def test_c3_full_size_several_views_against_reference_kernels(hip_device):
    cloud = synthetic.make_cloud(1_000_000, "band", 0)
    path = cameras.rotate360_path(1920, 1080, n_views=30)
    g = synthetic.upstream_grad(1080, 1920)
    for i in (0, 7, 19):
        hip, R, color, depth, radii, grads = _both(cloud, path[i], hip_device, g)
        assert R > 100_000
        _compare(hip, color, depth, radii, grads, 1_000_000, max_bad_pixels=16, label=f"C3 view {i} (num_rendered {R})")      # measured: 0-3


def test_c4_shape_against_reference_kernels(hip_device):
    cam, cloud = hp.box_setup(3_000_000, 2560, 1440)
    g = synthetic.upstream_grad(1440, 2560)
    hip, R, color, depth, radii, grads = _both(cloud, cam, hip_device, g)
    _compare(hip, color, depth, radii, grads, 3_000_000, max_bad_pixels=64, label=f"C4 shape (num_rendered {R})")      # measured: 8


def test_dist2_against_reference_kernels(hip_device):
    from simple_knn._C import distCUDA2
    pts = synthetic.make_cloud(300_000, "band", 5)["means3D"].to(hip_device)
    a, b = distCUDA2(pts).cpu().numpy(), ref_device.dist2(pts).cpu().numpy()
    assert np.array_equal(a, b) or np.abs(a - b).max() <= 1e-6 * b.max()
