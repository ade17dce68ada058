"""Shared test plumbing: run the same seeded inputs through the CPU oracle and through the HIP
rasterizer (via the public Python API -> C-ABI), and compare."""
import math

import numpy as np
import torch

from luciddreamer_amd import cameras, synthetic
from oracle import oracle

GRAD_NAMES = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")

# Stated tolerances (BASELINE.json north_star / SURVEY.md section 8c)
COLOR_ATOL = 1e-5          # max-abs RGB error vs oracle
DEPTH_RTOL = 1e-5          # depth: |d - d_ref| <= DEPTH_RTOL * max(1, |d_ref|)
GRAD_RTOL = 1e-4           # per tensor: max|g - g_ref| <= GRAD_RTOL * max|g_ref|
FRAGILE_FRAC = 2e-4        # at most this fraction of pixels may sit within 1 ulp-ish of a threshold flip


def tan_fov(cam):
    return math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)


def run_oracle(cloud, cam, degree, bg, grad_color=None, colors_precomp=None, cov3D_precomp=None,
               scale_modifier=1.0, use_sh=True, prefiltered=False):
    tfx, tfy = tan_fov(cam)
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    res = oracle.forward(
        n(bg), n(cloud["means3D"]), n(colors_precomp), n(cloud["opacities"]),
        None if cov3D_precomp is not None else n(cloud["scales"]),
        None if cov3D_precomp is not None else n(cloud["rotations"]),
        scale_modifier, n(cov3D_precomp), n(cam.world_view_transform), n(cam.full_proj_transform), tfx, tfy,
        cam.image_height, cam.image_width, n(cloud["shs"]) if (use_sh and colors_precomp is None) else None,
        degree, n(cam.camera_center), prefiltered)
    out = dict(color=res.color, depth=res.depth, radii=res.radii, num_rendered=res.num_rendered, res=res)
    if grad_color is not None:
        g = oracle.backward(res, n(grad_color))
        out["grads"] = dict(zip(GRAD_NAMES, g[:8]))
        out["grads"]["conic"] = g[8]
    return out


def run_hip(cloud, cam, degree, bg, device, grad_color=None, colors_precomp=None, cov3D_precomp=None,
            scale_modifier=1.0, use_sh=True, prefiltered=False, debug=False, grad_depth=None):
    """Through depth_diff_gaussian_rasterization_min's public API (autograd op -> _C -> C-ABI)."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    tfx, tfy = tan_fov(cam)
    d = lambda t: None if t is None else t.detach().to(device).requires_grad_(True)
    means3D, opac = d(cloud["means3D"]), d(cloud["opacities"])
    scales = None if cov3D_precomp is not None else d(cloud["scales"])
    rots = None if cov3D_precomp is not None else d(cloud["rotations"])
    cov = d(cov3D_precomp)
    shs = d(cloud["shs"]) if (use_sh and colors_precomp is None) else None
    cols = d(colors_precomp)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    cam_d = cam.to(device)
    rs = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=tfx, tanfovy=tfy, bg=bg.to(device),
        scale_modifier=scale_modifier, viewmatrix=cam_d.world_view_transform, projmatrix=cam_d.full_proj_transform,
        sh_degree=degree, campos=cam_d.camera_center, prefiltered=prefiltered, debug=debug)
    rast = GaussianRasterizer(raster_settings=rs)
    color, radii, depth = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, colors_precomp=cols,
                               scales=scales, rotations=rots, cov3D_precomp=cov)
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(),
               radii=radii.detach().cpu().numpy(), color_t=color, depth_t=depth)
    if grad_color is not None:
        loss = (color * grad_color.to(device)).sum()
        if grad_depth is not None:
            loss = loss + (depth * grad_depth.to(device)).sum()
        loss.backward()
        z = lambda t, shape: np.zeros(shape, np.float32) if (t is None or t.grad is None) else t.grad.cpu().numpy()
        P = means3D.shape[0]
        out["grads"] = dict(
            means2D=z(means2D, (P, 3)), colors=z(cols, (P, 3)), opacity=z(opac, (P, 1)), means3D=z(means3D, (P, 3)),
            cov3D=z(cov, (P, 6)), sh=z(shs, tuple(cloud["shs"].shape)), scales=z(scales, (P, 3)),
            rotations=z(rots, (P, 4)))
    return out


def compare_forward(hip, ref, check_exact=True, max_fragile=None):
    """Returns a dict of error figures; asserts the stated tolerances.  max_fragile: allowed number of pixels the
    oracle flags as sitting on a discrete threshold (default: max(2, FRAGILE_FRAC * pixels))."""
    st = ref["res"].stage()
    frag = st["fragile"]
    n_pix = frag.size
    frag_c = (frag & 1) != 0
    frag_d = (frag & 2) != 0
    allowed = max(8, FRAGILE_FRAC * n_pix) if max_fragile is None else max_fragile      # small images under heavy overdraw
    assert frag_c.sum() <= allowed, f"too many threshold-fragile pixels: {frag_c.sum()}"
    if check_exact:
        assert np.array_equal(hip["radii"], ref["radii"]), \
            f"radii differ at {np.nonzero(hip['radii'] != ref['radii'])[0][:10]}"
    cerr = np.abs(hip["color"] - ref["color"])
    cerr_ok = cerr[:, ~frag_c]
    derr = np.abs(hip["depth"][0] - ref["depth"][0]) / np.maximum(1.0, np.abs(ref["depth"][0]))
    derr_ok = derr[~(frag_c | frag_d)]
    # the error-bound band (bit 0) is wider than round 1's fixed band (bit 2).  Pixels that are EXEMPT only because
    # of the wider band and actually differ are counted and bounded: a regression confined to such pixels cannot hide there
    wide_only = frag_c & ((frag & 4) == 0)
    wide_only_failing = int(((cerr.max(axis=0) > COLOR_ATOL) & wide_only).sum())
    figures = dict(color_max=float(cerr_ok.max()) if cerr_ok.size else 0.0,
                   depth_rel_max=float(derr_ok.max()) if derr_ok.size else 0.0,
                   fragile_pixels=int(frag_c.sum()), fragile_color_max=float(cerr.max()),
                   fixed_band_pixels=int(((frag & 4) != 0).sum()), wide_band_only_failing=wide_only_failing)
    assert figures["color_max"] <= COLOR_ATOL, figures
    assert figures["depth_rel_max"] <= DEPTH_RTOL, figures
    assert wide_only_failing <= max(4, 4e-6 * n_pix), figures
    return figures


def compare_grads(hip_g, ref_g, names=None, rtol=GRAD_RTOL):
    figures = {}
    for k in (names or GRAD_NAMES):
        a, b = hip_g[k], ref_g[k]
        a = a.reshape(b.shape)
        scale = float(np.abs(b).max())
        err = float(np.abs(a - b).max())
        figures[k] = (err, scale)
        if scale == 0.0:
            assert err == 0.0, f"{k}: reference gradient is identically zero but HIP gave {err}"
        else:
            assert err <= rtol * scale, f"{k}: max|g-g_ref|={err:.3e} > {rtol}*max|g_ref|={scale:.3e}"
    return figures


def compare_grads_by_row(hip, ref, P, names=("means2D", "opacity", "means3D", "sh", "scales", "rotations"), max_outliers=32):
    """The full-size bar: every gradient row within GRAD_RTOL of its tensor's maximum, except rows of Gaussians whose
    footprint contains a pixel the oracle itself flags as sitting within rounding of a discrete threshold (alpha = 1/255,
    T = 1e-4: "fragile") -- those within 1e-3, at most `max_outliers` per tensor, and each one must be located on such
    a pixel.  Returns {tensor: (worst relative row error, outliers)}."""
    st = ref["res"].stage()
    fy, fx = np.nonzero(st["fragile"] != 0)
    m2, radii = st["means2D"], ref["radii"]
    report = {}
    for k in names:
        a = hip["grads"][k].reshape(P, -1)
        b = ref["grads"][k].reshape(P, -1)
        scale = float(np.abs(b).max())
        row_err = np.abs(a - b).max(axis=1)
        bad = np.nonzero(row_err > GRAD_RTOL * scale)[0]
        report[k] = (f"{row_err.max() / scale:.2e}", len(bad))
        assert len(bad) <= max_outliers, (k, len(bad))
        assert row_err.max() <= 1e-3 * scale, (k, row_err.max(), scale)
        for i in bad:
            # each outlier must TOUCH a flagged pixel: its own alpha there reaches the 1/255 threshold (to within 10 %),
            # i.e. the pixel lies inside the splat's actual support, not merely inside its bounding square
            ca, cb, cc, op = st["conic_opacity"][i].astype(np.float64)
            dx, dy = m2[i, 0] - fx.astype(np.float64), m2[i, 1] - fy.astype(np.float64)
            power = -0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy
            touches = (power <= 1e-6) & (op * np.exp(np.minimum(power, 0.0)) >= 0.9 / 255.0)
            assert touches.any(), f"{k}: Gaussian {i} differs by {row_err[i] / scale:.2e} and touches no threshold-fragile pixel"
    return report


def box_setup(P, W, H, seed=0, scale_mult=1.0, sh_coeffs=16):
    cam = cameras.identity_camera(W, H)
    cloud = synthetic.make_cloud(P, "box", seed, sh_coeffs=sh_coeffs, scale_mult=scale_mult)
    return cam, cloud
