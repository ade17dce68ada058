"""GPU: how much does the REFERENCE disagree with ITSELF?  The budget of the exemptions the full-size parity tests use.

The blend is a discontinuous function of its float32 inputs (alpha >= 1/255, power <= 0, T * (1 - alpha) >= 1e-4:
forward.cu:331-347, backward.cu:500-515): two correct float32 evaluations -- different FMA contraction, a different exp --
flip those decisions on a handful of pixels, and a flipped pixel moves the image by up to 1e-3 and the gradient rows of
the splats on it by more than 1e-4 of the tensor's maximum.  The only defensible budget for such pixels / rows is the
count by which the reference's own code differs between its own builds, on the same C3 views and the C4 shape:

  host     oracle/_ref/libref_raster.so             its .cu sources compiled for the host by g++, -ffp-contract=off
  gfx950   oracle/_ref/libref_raster_gfx950.so      the same sources by hipcc for this GPU, -ffp-contract=off (the checker)
  fma      oracle/_ref/libref_raster_gfx950_fma.so  the same, with the compiler's DEFAULT contraction (-ffp-contract=fast;
                                                    nvcc's default -fmad=true makes the same choice: the reference as its
                                                    own setup.py builds it)

host and gfx950 evaluate the same IEEE operations in the same order and differ only in their exp(): measured, they agree on
every discrete decision of all three C3 views (0 pixels beyond 1e-5; C4 shape: 1).  fma against gfx950 is the reference
under its own default build flags against its strict build.  Measured on MI355X (profiles/parity_calibration.json): C3
view 0 -- the identity camera, where contraction changes nothing -- 0 pixels; view 11: 4112 pixels and 11-48 gradient rows
per tensor; view 19: 6646 pixels and 15-39 rows; C4 shape (identity camera): 14 pixels, radii differ.  The HIP path against
the strict build: 3 / 0 / 0 pixels, 0-2 rows (C3); 13 pixels, 0-1 rows (C4 shape).  So the budgets the full-size tests
allow -- 16 pixels per C3 view, 64 at the C4 shape, 8 rows per tensor up to 1.5e-3 -- are two orders of magnitude below what
the reference's own default build moves by on a generic view.  This file measures all of it, writes
gpurun_out/parity_calibration.json (bench.py quotes the committed copy under profiles/), and asserts the absolute budgets
per case and, over the three C3 views together, that the HIP path disagrees with the strict reference on fewer pixels and
rows than the reference's default build does."""
import json
import os

import numpy as np
import pytest
import torch

from luciddreamer_amd import cameras, synthetic
from oracle import ref, ref_device
from tests import helpers as hp

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (ref_device.available() and ref.available()),
                                                  reason="oracle/_ref builds did not travel")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "parity_calibration.json")
TENSORS = ("means2D", "opacity", "means3D", "sh", "scales", "rotations")


def _host_reference(cloud, cam, bg, g, degree=3):
    n = lambda t: t.detach().cpu().numpy()
    tfx, tfy = hp.tan_fov(cam)
    res = ref.forward(n(bg), n(cloud["means3D"]), None, n(cloud["opacities"]), n(cloud["scales"]), n(cloud["rotations"]), 1.0,
                      None, n(cam.world_view_transform), n(cam.full_proj_transform), tfx, tfy, cam.image_height,
                      cam.image_width, n(cloud["shs"]), degree, n(cam.camera_center), False)
    grads = dict(zip(hp.GRAD_NAMES, ref.backward(res, n(g))[:8]))
    return dict(color=res.color, depth=res.depth, radii=res.radii, grads=grads)


def _device_reference(cloud, cam, bg, g, dev, degree=3, contract="off"):
    c = {k: v.to(dev).contiguous() for k, v in cloud.items()}
    cd = cam.to(dev)
    r = ref_device.Renderer(contract)
    tfx, tfy = hp.tan_fov(cam)
    R, color, depth, radii = r.forward(bg.to(dev), c["means3D"], None, c["opacities"], c["scales"], c["rotations"], 1.0, None,
                                       cd.world_view_transform.contiguous(), cd.full_proj_transform.contiguous(), tfx, tfy,
                                       cam.image_height, cam.image_width, c["shs"], degree, cd.camera_center.contiguous())
    grads = r.backward(g.to(dev).contiguous())
    names = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
    return dict(color=color.cpu().numpy(), depth=depth.cpu().numpy(), radii=radii.cpu().numpy(),
                grads={k: t.cpu().numpy() for k, t in zip(names, grads)}, num_rendered=int(R))


def disagreement(a, b, P):
    """Pixels beyond the stated image tolerance and gradient rows beyond the stated gradient tolerance, `a` against `b`."""
    cerr = np.abs(a["color"] - b["color"]).max(axis=0)
    derr = np.abs(a["depth"][0] - b["depth"][0]) / np.maximum(1.0, np.abs(b["depth"][0]))
    out = {"pixels_beyond_1e-5": int(((cerr > hp.COLOR_ATOL) | (derr > hp.DEPTH_RTOL)).sum()),
           "max_abs_rgb": float(cerr.max()), "radii_equal": bool(np.array_equal(a["radii"], b["radii"])), "rows_beyond_1e-4": {},
           "worst_row_rel": {}}
    for k in TENSORS:
        x, y = a["grads"][k].reshape(P, -1), b["grads"][k].reshape(P, -1)
        scale = float(np.abs(y).max())
        row = np.abs(x - y).max(axis=1)
        out["rows_beyond_1e-4"][k] = int((row > hp.GRAD_RTOL * scale).sum())
        out["worst_row_rel"][k] = float(row.max() / scale) if scale > 0 else 0.0
    return out


def _case(label, cloud, cam, dev, seed=0):
    P = cloud["means3D"].shape[0]
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(cam.image_height, cam.image_width, seed=seed)
    host = _host_reference(cloud, cam, bg, g)
    devr = _device_reference(cloud, cam, bg, g, dev)
    fma = _device_reference(cloud, cam, bg, g, dev, contract="fast") if ref_device.available("fast") else None
    hip = hp.run_hip(cloud, cam, 3, bg, dev, g)
    rec = {"case": label, "num_rendered": devr["num_rendered"], "pixels": int(cam.image_height * cam.image_width),
           "reference_host_vs_reference_gfx950": disagreement(host, devr, P),
           "reference_fma_vs_reference_gfx950": disagreement(fma, devr, P) if fma is not None else None,
           "hip_vs_reference_gfx950": disagreement(hip, devr, P),
           "hip_vs_reference_host": disagreement(hip, host, P)}
    print(json.dumps(rec))
    return rec


def _within(rec, max_pixels):
    self_, ours = rec["reference_fma_vs_reference_gfx950"], rec["hip_vs_reference_gfx950"]
    assert ours["radii_equal"] and rec["hip_vs_reference_host"]["radii_equal"], rec["case"]
    assert rec["reference_host_vs_reference_gfx950"]["radii_equal"], rec["case"]
    assert ours["pixels_beyond_1e-5"] <= max_pixels, rec
    for k in TENSORS:
        assert ours["rows_beyond_1e-4"][k] <= 8 and ours["worst_row_rel"][k] <= 1.5e-3, (rec["case"], k, rec)


def _within_reference_self_disagreement(recs):
    """Over several generic views together: fewer disagreements with the strict reference than its default build has."""
    if any(r["reference_fma_vs_reference_gfx950"] is None for r in recs):
        return
    tot = lambda pair, f: sum(f(r[pair]) for r in recs)
    assert tot("hip_vs_reference_gfx950", lambda x: x["pixels_beyond_1e-5"]) <= tot("reference_fma_vs_reference_gfx950", lambda x: x["pixels_beyond_1e-5"])
    for k in TENSORS:
        assert tot("hip_vs_reference_gfx950", lambda x: x["rows_beyond_1e-4"][k]) <= tot("reference_fma_vs_reference_gfx950", lambda x: x["rows_beyond_1e-4"][k]), k


def _store(recs):
    from luciddreamer_amd import _lib
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    old = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
    old["lr_version"] = _lib.lib().lr_version().decode()
    old["what"] = ("pixels beyond 1e-5 (colour max-abs or relative depth) and gradient rows beyond 1e-4 of the tensor's maximum: the "
                   "reference's host build and its default-contraction gfx950 build against its strict gfx950 build (the "
                   "reference's disagreement with itself), and the HIP path against the strict builds; tests/test_gpu_ref_selfcal.py")
    for r in recs:
        old["cases"][r["case"]] = r
    json.dump(old, open(OUT, "w"), indent=1)


def test_c3_views_reference_self_disagreement_bounds_ours(hip_device):
    cloud = synthetic.make_cloud(1_000_000, "band", 0)
    path = cameras.rotate360_path(1920, 1080, n_views=30)
    recs = [_case(f"c3_view{i}", cloud, path[i], hip_device) for i in (0, 11, 19)]
    _store(recs)
    for r in recs:
        _within(r, 16)
    _within_reference_self_disagreement(recs)


def test_strict_mode_agrees_with_the_strict_reference_on_every_pixel(hip_device):
    """config.set_strict_parity(True): the blend evaluates alpha with the reference's own float operations (no contraction,
    expf), so every discrete decision falls as in the reference's strict build: NO pixel beyond 1e-5 and NO gradient row
    beyond 1e-4 on the C3 views -- including view 0, where the default mode differs on 3 pixels -- and the numbers go into the
    calibration file next to the default mode's."""
    from luciddreamer_amd import config
    cloud = synthetic.make_cloud(1_000_000, "band", 0)
    path = cameras.rotate360_path(1920, 1080, n_views=30)
    P = 1_000_000
    bg = torch.zeros(3)
    recs = []
    config.set_strict_parity(True)
    try:
        for i in (0, 11, 15, 19):
            g = synthetic.upstream_grad(1080, 1920)
            devr = _device_reference(cloud, path[i], bg, g, hip_device)
            hip = hp.run_hip(cloud, path[i], 3, bg, hip_device, g)
            d = disagreement(hip, devr, P)
            print(f"strict mode, C3 view {i}:", json.dumps(d))
            recs.append((i, d))
    finally:
        config.set_strict_parity(False)
    old = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
    old["strict_mode_hip_vs_reference_gfx950"] = {f"c3_view{i}": d for i, d in recs}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(old, open(OUT, "w"), indent=1)
    for i, d in recs:
        assert d["radii_equal"] and d["pixels_beyond_1e-5"] == 0, (i, d)
        assert all(v == 0 for v in d["rows_beyond_1e-4"].values()), (i, d)


def test_c4_shape_reference_self_disagreement_bounds_ours(hip_device):
    cam, cloud = hp.box_setup(3_000_000, 2560, 1440)
    rec = _case("c4_shape", cloud, cam, hip_device)
    _store([rec])
    _within(rec, 64)
    # the same in strict mode: no pixel, no row
    from luciddreamer_amd import config
    bg, g = torch.zeros(3), synthetic.upstream_grad(cam.image_height, cam.image_width, seed=0)
    devr = _device_reference(cloud, cam, bg, g, hip_device)
    config.set_strict_parity(True)
    try:
        hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    finally:
        config.set_strict_parity(False)
    d = disagreement(hip, devr, 3_000_000)
    print("strict mode, C4 shape:", json.dumps(d))
    old = json.load(open(OUT))
    old.setdefault("strict_mode_hip_vs_reference_gfx950", {})["c4_shape"] = d
    json.dump(old, open(OUT, "w"), indent=1)
    assert d["radii_equal"] and d["pixels_beyond_1e-5"] == 0 and all(v == 0 for v in d["rows_beyond_1e-4"].values()), d
