"""GPU: the HIP path against the REFERENCE'S OWN rasterizer code -- (i) the committed outputs of oracle/_ref
(tests/golden/ref_raster_fixtures.npz, made by tests/golden/make_ref_fixtures.py from the reference's .cu sources compiled
for the host) on the seeded edge cases of tests/ref_cases.py, and (ii) oracle/_ref run live on the GPU box's host cores
at BASELINE.json configs[1] size (100 k Gaussians, SH 3, 1080p).  Tolerances: image max-abs <= 1e-5, depth 1e-5 relative,
gradients 1e-4 of each tensor's maximum."""
import os

import numpy as np
import pytest
import torch

from luciddreamer_amd import synthetic
from oracle import oracle, ref
from tests import helpers as hp, ref_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_raster_fixtures.npz")
CASES = {c["name"]: c for c in ref_cases.all_cases()}
FIXTURE_CASES = ("box_deg3", "near_plane", "fov_clamp", "depth_ties", "sh_clamp", "needles", "cov3D_precomp", "single", "M1")
PAIRS = (("means2D", "means2D"), ("colors", "colors"), ("opacity", "opacity"), ("means3D", "means3D"), ("cov3D", "cov3D"),
         ("sh", "sh"), ("scales", "scales"), ("rotations", "rotations"))


def _run_hip_case(c, dev):
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True)
    use_cov, use_sh = c["cov3D_precomp"] is not None, c["colors_precomp"] is None
    means3D, opac = t(c["means3D"]), t(c["opacities"])
    scales, rots = (None, None) if use_cov else (t(c["scales"]), t(c["rotations"]))
    cov, shs, cols = t(c["cov3D_precomp"]), (t(c["shs"]) if use_sh else None), t(c["colors_precomp"])
    means2D = torch.zeros_like(means3D, requires_grad=True)
    f = lambda a: torch.from_numpy(a).to(dev)
    rs = GaussianRasterizationSettings(c["H"], c["W"], c["tanfovx"], c["tanfovy"], f(c["bg"]), c["scale_modifier"],
                                       f(c["view"]), f(c["proj"]), c["degree"], f(c["campos"]), False, False)
    color, radii, depth = GaussianRasterizer(rs)(means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                                 colors_precomp=cols, scales=scales, rotations=rots, cov3D_precomp=cov)
    (color * f(c["dL_dcolor"])).sum().backward()
    P = means3D.shape[0]
    z = lambda x, shape: np.zeros(shape, np.float32) if (x is None or x.grad is None) else x.grad.cpu().numpy()
    grads = dict(means2D=z(means2D, (P, 3)), colors=z(cols, (P, 3)), opacity=z(opac, (P, 1)), means3D=z(means3D, (P, 3)),
                 cov3D=z(cov, (P, 6)), sh=z(shs, c["shs"].shape), scales=z(scales, (P, 3)), rotations=z(rots, (P, 4)))
    return color.detach().cpu().numpy(), depth.detach().cpu().numpy(), radii.cpu().numpy(), grads


@pytest.mark.parametrize("name", FIXTURE_CASES)
def test_hip_reproduces_committed_reference_outputs(hip_device, name):
    fx = np.load(GOLD)
    c = CASES[name]
    color, depth, radii, grads = _run_hip_case(c, hip_device)
    assert np.array_equal(radii, fx[name + "/radii"])
    if name == "needles":
        # near-singular conics: the exponent is a sum of products 1e4 times its own size, no two float32 evaluations agree
        # to 1e-5 (the reference's own builds do not either) -- judged against a float64 evaluation in
        # test_ill_conditioned_splats_against_the_float64_blend below
        assert np.isfinite(color).all() and np.isfinite(depth).all() and all(np.isfinite(g).all() for g in grads.values())
        # a BOUNDED agreement with the committed reference image all the same -- a blend regression confined to
        # high-overdraw pixels must not pass unseen: most pixels agree to the usual tolerance, the rest stay close
        cerr = np.abs(color - fx[name + "/color"]).max(axis=0)
        print(f"needles vs the committed reference image: median {np.median(cerr):.2e}, pixels beyond 1e-5: "
              f"{(cerr > 1e-5).mean():.3f}, beyond 1e-3: {(cerr > 1e-3).mean():.4f}, max {cerr.max():.2e}")
        # measured on MI355X: median 1.25e-5, 55 % of the pixels beyond 1e-5 (no two float32 evaluations of these conics agree
        # there), 0.18 % beyond 1e-3, one pixel off by a whole layer
        assert np.median(cerr) <= 5e-5 and (cerr > 1e-3).mean() <= 0.01 and np.percentile(cerr, 99) <= 1e-3
        return
    # pixels where the reference itself sits within an ulp of a discrete threshold: flagged by the (bit-identical)
    # restatement, which records them while blending
    frag = oracle.forward(*ref_cases.forward_args(c)).stage()["fragile"]
    ok = (frag & 1) == 0
    # (these cases are small images under heavy overdraw -- up to ~700 pairs per pixel -- so a handful is expected)
    assert (~ok).sum() <= max(16, hp.FRAGILE_FRAC * ok.size), int((~ok).sum())
    assert np.abs(color - fx[name + "/color"])[:, ok].max() <= hp.COLOR_ATOL
    ok_d = ok & ((frag & 2) == 0)
    rd = fx[name + "/depth"][0]
    assert (np.abs(depth[0] - rd) / np.maximum(1.0, np.abs(rd)))[ok_d].max() <= hp.DEPTH_RTOL
    rtol = hp.GRAD_RTOL
    for k, kr in PAIRS:
        b = fx[f"{name}/dL_d{kr}"]
        if c["cov3D_precomp"] is not None and k in ("scales", "rotations"):
            continue
        if c["cov3D_precomp"] is None and k == "cov3D":
            continue                  # an intermediate of the reference's backward, not an autograd output here
        if (c["colors_precomp"] is None) == (k == "colors"):
            continue
        a = grads[k].reshape(b.shape)
        scale = float(np.abs(b).max())
        assert np.abs(a - b).max() <= rtol * scale + 1e-30, (k, float(np.abs(a - b).max()), scale)


def _sliver_case():
    """A second ill-conditioned scene (not among the committed fixtures: the reference is evaluated live): splats 200 x longer
    than wide close to the camera, conic entries of 1e4..1e8."""
    rng = np.random.default_rng(61)
    cl = ref_cases._cloud(rng, 600, (-0.5, -0.35, 0.35), (0.5, 0.35, 1.6), 0.02)
    cl["scales"][:, 0] *= np.float32(60.0)
    cl["scales"][:, 1:] *= np.float32(3e-3)
    return ref_cases._case("slivers", ref_cases._cam(128, 80), cl, degree=2)


def _reference_on_device(c, dev):
    from oracle import ref_device
    if not ref_device.available():
        return None
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r = ref_device.Renderer()
    _, color, _, _ = r.forward(t(c["bg"]), t(c["means3D"]), t(c["colors_precomp"]), t(c["opacities"]), t(c["scales"]),
                               t(c["rotations"]), c["scale_modifier"], t(c["cov3D_precomp"]), t(c["view"]), t(c["proj"]),
                               c["tanfovx"], c["tanfovy"], c["H"], c["W"], t(c["shs"]) if c["colors_precomp"] is None else None,
                               c["degree"], t(c["campos"]))
    return color.cpu().numpy()


@pytest.mark.parametrize("name", ["needles", "slivers"])
def test_ill_conditioned_splats_against_the_float64_blend(hip_device, name):
    """Needle splats: conic entries of 1e4..1e9 whose quadratic form cancels to O(1) in the exponent, so two correct float32
    evaluations (the reference's `-0.5f*(a dx dx + c dy dy) - b dx dy`, this library's Horner form on scaled coefficients,
    the reference compiled with or without FMA contraction) differ by the rounding residue of their operand order -- no
    float32 image is "the" answer.  The yardstick is the same blend evaluated in float64 on the same float32 records and
    tile lists (oracle.blend_f64).  Measured against it: the HIP image, the reference's host build (= the restatement, bit
    for bit) and the reference's kernels compiled for this GPU.  The HIP image has to be as close to the float64 image as
    the reference's own builds are."""
    c = CASES[name] if name in CASES else _sliver_case()
    color, depth, radii, grads = _run_hip_case(c, hip_device)
    o = oracle.forward(*ref_cases.forward_args(c))
    assert np.array_equal(radii, o.radii)                                  # the per-Gaussian stage is exact all the same
    assert np.isfinite(color).all() and np.isfinite(depth).all() and all(np.isfinite(g).all() for g in grads.values())
    if name == "needles":                                                  # the restatement IS the reference here
        fx = np.load(GOLD)
        assert np.array_equal(o.color.view(np.uint32), fx["needles/color"].view(np.uint32))
    f64 = oracle.blend_f64(o, c["bg"], c["colors_precomp"])
    images = {"hip": color, "reference_host": o.color}
    dev_img = _reference_on_device(c, hip_device)
    if dev_img is not None:
        images["reference_gfx950"] = dev_img
    # pixels where the reference's own value is within ITS float32 error bound of a discrete decision (sign of the exponent,
    # alpha = 1/255, T = 1e-4): any evaluation may take the other branch there and be off by a whole layer
    frag = (o.stage()["fragile"] & 1) != 0
    assert frag.mean() <= 0.10, float(frag.mean())
    stats = {}
    for k, img in images.items():
        e = np.abs(img.astype(np.float64) - f64).max(axis=0)
        ok = e[~frag]
        stats[k] = dict(max=float(ok.max()), p99=float(np.quantile(ok, 0.99)), mean=float(ok.mean()),
                        within_1e5=float((ok <= 1e-5).mean()), max_on_flagged=float(e[frag].max()) if frag.any() else 0.0)
    print(name, f"{int(frag.sum())} of {frag.size} pixels flagged;",
          {k: {a: (f"{b:.2e}" if a != "within_1e5" else f"{100 * b:.1f}%") for a, b in v.items()} for k, v in stats.items()})
    refs = [v for k, v in stats.items() if k != "hip"]
    best_ref = {a: min(r[a] for r in refs) for a in ("max", "p99", "mean")}
    h = stats["hip"]
    # the HIP image is as close to the float64 image as the reference's own builds are (10 % slack on the statistics)
    assert h["mean"] <= 1.1 * best_ref["mean"] + 1e-7, (h, best_ref)
    assert h["p99"] <= 1.1 * best_ref["p99"] + 1e-6, (h, best_ref)
    assert h["max"] <= 1.1 * best_ref["max"] + 1e-5, (h, best_ref)
    assert h["within_1e5"] >= max(r["within_1e5"] for r in refs) - 0.01, (h, refs)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libref_raster.so did not travel")
def test_c2_size_against_compiled_reference(hip_device):
    """100 k Gaussians, SH 3, 1080p, forward+backward: HIP vs the reference's own kernels on the host cores."""
    cam, cloud = hp.box_setup(100_000, 1920, 1080)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(1080, 1920)
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    tfx, tfy = hp.tan_fov(cam)
    r = ref.forward(n(bg), n(cloud["means3D"]), None, n(cloud["opacities"]), n(cloud["scales"]), n(cloud["rotations"]), 1.0,
                    None, n(cam.world_view_transform), n(cam.full_proj_transform), tfx, tfy, 1080, 1920, n(cloud["shs"]), 3,
                    n(cam.camera_center))
    gr = ref.backward(r, n(g))
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    o = hp.run_oracle(cloud, cam, 3, bg)                       # for the fragile-pixel flags only
    assert np.array_equal(o["color"].view(np.uint32), r.color.view(np.uint32)), "restatement != compiled reference"
    refd = dict(color=r.color, depth=r.depth, radii=r.radii, res=o["res"])
    fig = hp.compare_forward(hip, refd)
    refd["grads"] = dict(zip(hp.GRAD_NAMES, gr[:8]))
    report = hp.compare_grads_by_row(hip, refd, 100_000, max_outliers=4)       # the bar of test_gpu_full.py's C2 / C3
    print("C2 vs oracle/_ref:", r.num_rendered, fig, report)
