"""GPU tests at BASELINE.json sizes, against the committed fixture, at stage level, and through
size-independent properties where the oracle would be too slow."""
import math
import os

import numpy as np
import pytest
import torch

from luciddreamer_amd import cameras, synthetic
from tests import helpers as hp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _raw_forward(cloud, cam, degree, bg, dev, binning_capacity=0):
    """Direct _C call so the opaque scratch buffers can be inspected."""
    from luciddreamer_amd import _C
    tfx, tfy = hp.tan_fov(cam)
    c = cam.to(dev)
    e = torch.Tensor([])
    return _C.rasterize_gaussians(bg.to(dev), cloud["means3D"].to(dev), e, cloud["opacities"].to(dev),
                                  cloud["scales"].to(dev), cloud["rotations"].to(dev), 1.0, e, c.world_view_transform,
                                  c.full_proj_transform, tfx, tfy, cam.image_height, cam.image_width,
                                  cloud["shs"].to(dev), degree, c.camera_center, False, False,
                                  binning_capacity=binning_capacity)


def _align(n):
    return (n + 255) // 256 * 256


def _unpack(out, P, W, H):
    """Mirror of csrc/common.h geom/img/bin layouts (opaque to users; the test knows them)."""
    num_rendered, color, depth, radii, geom, binning, img = out
    g = geom.cpu().numpy()
    hdr = g[:32].view(np.uint32)          # num_rendered, overflow, trap, capacity, P, num_sorted, num_instances, bin_bound
    rec = g[256:256 + 48 * P].view(np.float32).reshape(P, 12)
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    im = img.cpu().numpy()
    final_T = im[:4 * N].view(np.float32).reshape(H, W)
    n_contrib = im[_align(4 * N):_align(4 * N) + 4 * N].view(np.uint32).reshape(H, W)
    ranges = im[2 * _align(4 * N):2 * _align(4 * N) + 8 * T].view(np.uint32).reshape(T, 2)
    n_inst = int(hdr[5])
    b = binning.cpu().numpy()
    emission = b[:4 * n_inst].view(np.uint32)               # final tile-sorted list (emission slots), offset 0
    Rb = max(int(hdr[7]), 1)                                # hdr[7] = bin_bound the layout was computed for
    gid_off = _align(4 * Rb) + _align(8 * Rb)               # point_list, 64-bit sort words, then inst_gid
    inst_gid = b[gid_off:gid_off + 4 * Rb].view(np.uint32)
    point_list = inst_gid[emission]                         # Gaussian index of every list entry
    return dict(rec=rec, final_T=final_T, n_contrib=n_contrib, ranges=ranges, point_list=point_list, hdr=hdr)


def _max_alpha_on_tile(st, gid, tile, gx):
    """Oracle-side check of a dropped instance: its alpha on every pixel of the tile (float64)."""
    tx, ty = tile % gx, tile // gx
    xs, ys = np.meshgrid(np.arange(tx * 16, tx * 16 + 16), np.arange(ty * 16, ty * 16 + 16))
    dx = st["means2D"][gid, 0].astype(np.float64) - xs
    dy = st["means2D"][gid, 1].astype(np.float64) - ys
    a, b, c, o = st["conic_opacity"][gid].astype(np.float64)
    power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
    return float((o * np.exp(np.minimum(power, 0.0))).max())


@pytest.mark.parametrize("P,W,H,scale_mult,flat_depth", [
    (20_000, 320, 192, 1.5, False),      # ~50 instances per tile: the one-wave sort
    (150_000, 160, 96, 1.0, False),      # thousands per tile: the bucket sort / the 128 KB class
    (30_000, 256, 256, 1.5, True),       # every splat at ONE depth: ties resolved by index, bucket pile-up -> network fallback
    (400_000, 64, 64, 0.7, False),       # > 16384 per tile: sorted in place in global memory
])
def test_stage_outputs_are_bit_exact(hip_device, P, W, H, scale_mult, flat_depth):
    """Per-Gaussian records are bit-identical to the oracle's; the reported num_rendered is the reference's;
    every per-tile list is the oracle's list, in the oracle's order, minus instances that exact tile culling
    dropped -- and every dropped instance provably contributes to no pixel of its tile.  The size classes of the
    per-tile sort (tilebin.hip) are all exercised."""
    cam, cloud = hp.box_setup(P, W, H, scale_mult=scale_mult)
    if flat_depth:
        cloud["means3D"][:, 2] = 4.0
    bg = torch.zeros(3)
    ref = hp.run_oracle(cloud, cam, 3, bg)
    st = ref["res"].stage()
    out = _raw_forward(cloud, cam, 3, bg, hip_device)
    assert out[0] == ref["num_rendered"]
    u = _unpack(out, P, W, H)
    per_tile = (u["ranges"][:, 1].astype(np.int64) - u["ranges"][:, 0]).max()
    print(f"largest tile list: {per_tile}")
    assert int(u["hdr"][0]) == ref["num_rendered"]
    vis = ref["radii"] > 0
    rec = u["rec"][vis]
    assert np.array_equal(rec[:, 0:2], st["means2D"][vis])
    assert np.array_equal(rec[:, 2:4], st["conic_opacity"][vis][:, 0:2])
    assert np.array_equal(rec[:, 4], st["conic_opacity"][vis][:, 2])
    assert np.array_equal(rec[:, 5], st["conic_opacity"][vis][:, 3])
    assert np.array_equal(rec[:, 6:9], st["rgb"][vis])
    assert np.array_equal(rec[:, 9], st["depths"][vis])
    n_inst = u["point_list"].shape[0]
    assert 0 < n_inst <= ref["num_rendered"]
    rng, orng = u["ranges"].astype(np.int64), st["ranges"].astype(np.int64)
    assert int((rng[:, 1] - rng[:, 0]).sum()) == n_inst
    gx = (W + 15) // 16
    dropped_checked = 0
    for t in range(rng.shape[0]):
        ours = u["point_list"][rng[t, 0]:rng[t, 1]]
        theirs = st["point_list"][orng[t, 0]:orng[t, 1]]
        keep = np.isin(theirs, ours)
        assert np.array_equal(theirs[keep], ours), f"tile {t}: not an order-preserving sub-list of the reference list"
        for gid in theirs[~keep][:3]:
            assert _max_alpha_on_tile(st, gid, t, gx) < 1.0 / 255.0
            dropped_checked += 1
    print(f"instances: reference {ref['num_rendered']}, after exact tile culling {n_inst}; checked {dropped_checked} dropped")
    frag = (st["fragile"] & 1) != 0
    assert np.abs(u["final_T"] - st["final_T"])[~frag].max() <= 2e-6


def test_against_committed_fixture(hip_device):
    from tests.golden.make_oracle_fixture import SPEC
    fix = np.load(os.path.join(GOLD, "oracle_e2e_fixture.npz"))
    cam, cloud = hp.box_setup(SPEC["P"], SPEC["W"], SPEC["H"], seed=SPEC["seed"], scale_mult=1.5)
    g = synthetic.upstream_grad(SPEC["H"], SPEC["W"], seed=SPEC["grad_seed"])
    hip = hp.run_hip(cloud, cam, SPEC["degree"], torch.tensor(SPEC["bg"]), hip_device, g)
    assert np.array_equal(hip["radii"], fix["radii"])
    ok = (fix["fragile"] & 1) == 0
    assert np.abs(hip["color"] - fix["color"])[:, ok].max() <= hp.COLOR_ATOL
    ok_d = ok & ((fix["fragile"] & 2) == 0)
    assert (np.abs(hip["depth"][0] - fix["depth"][0]) / np.maximum(1, np.abs(fix["depth"][0])))[ok_d].max() <= hp.DEPTH_RTOL
    for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"):
        a, b = hip["grads"][k].reshape(fix["grad_" + k].shape), fix["grad_" + k]
        assert np.abs(a - b).max() <= hp.GRAD_RTOL * np.abs(b).max(), k


def test_c2_full_size_parity(hip_device):
    """BASELINE.json configs[1]: 100k Gaussians, SH degree 3, 1080p, forward+backward vs the oracle."""
    cam, cloud = hp.box_setup(100_000, 1920, 1080)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(1080, 1920)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    fig = hp.compare_forward(hip, ref)
    # the same bar as C3 below: rows within 1e-4 of the tensor's maximum; a row beyond it must belong to a Gaussian with
    # an oracle-flagged threshold pixel in its footprint (and stay within 1e-3)
    report = hp.compare_grads_by_row(hip, ref, 100_000, max_outliers=4)
    print("C2", ref["num_rendered"], fig, report)


def test_c3_full_size_parity_one_view(hip_device):
    """BASELINE.json configs[2] (the metric's configuration): 1e6 Gaussians, SH degree 3, 1080p, band cloud, one view
    of the rotate360 path, forward+backward vs the oracle at full size.

    With ~3e7 pixel-Gaussian pairs a handful of pixels sit within an ulp of a discrete threshold (alpha = 1/255,
    T = 1e-4): the oracle flags them ("fragile"), the images are compared outside them, and a Gaussian's gradient may
    differ by the contribution of such a pixel.  Stated bar: every gradient row within 1e-4 of the tensor's max,
    except rows of Gaussians whose 3-sigma footprint contains a flagged pixel -- those within 1e-3, and few."""
    cloud = synthetic.make_cloud(1_000_000, "band", 0)
    cam = cameras.rotate360_path(1920, 1080, n_views=30)[11]
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(1080, 1920)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    fig = hp.compare_forward(hip, ref)
    report = hp.compare_grads_by_row(hip, ref, 1_000_000)
    print("C3", ref["num_rendered"], fig, report)


@pytest.mark.parametrize("strict", [False, True])
def test_c3_headline_step_three_views_in_flight_vs_oracle(hip_device, strict):
    """The HEADLINE's own entry point and kernels at the metric's size (BASELINE.json configs[2]: 1e6 Gaussians, 1080p, band
    cloud, rotate360 path): ONE lr_views_accumulate call over views 0 / 11 / 19 with three views in flight -- which at 1080p
    launches k_render_fwd_tile + k_render_bwd_tile (asserted) -- against the SUM of the oracle's three backwards
    (backward.cu:399-586 per view; gradients are additive over views), and each view's image from the same kernels (the drop-in
    operator under the views_in_flight hint) against the oracle's image (forward.cu:261-391).  Default mode: the bar of
    test_c3_full_size_parity_one_view (rows within 1e-4 of the tensor's maximum; the few beyond must touch an oracle-flagged
    threshold pixel of one of the views, within 1.5e-3).  Strict mode: no pixel and no row exempt."""
    from luciddreamer_amd import _lib, config, parallel
    P, W, H = 1_000_000, 1920, 1080
    view_ids = (0, 11, 19)
    cloud = synthetic.make_cloud(P, "band", 0)
    path = cameras.rotate360_path(W, H, n_views=30)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(H, W)
    names = ("means2D", "opacity", "means3D", "sh", "scales", "rotations")
    config.reset()
    config.set_strict_parity(strict)
    try:
        total, flagged, worst_R = {}, [], 0
        for vi in view_ids:
            ref = hp.run_oracle(cloud, path[vi], 3, bg, g)
            worst_R = max(worst_R, ref["num_rendered"])
            # image, depth, radii of THIS view from the kernels the headline runs
            _lib.tune_set("views_in_flight", 3)
            try:
                hip = hp.run_hip(cloud, path[vi], 3, bg, hip_device, g)
                assert _lib.last_launch_shapes() == ("tile", "tile")
            finally:
                _lib.tune_set("views_in_flight", -1)
            st = ref["res"].stage()
            if strict:
                assert np.array_equal(hip["radii"], ref["radii"])
                assert np.abs(hip["color"] - ref["color"]).max() <= hp.COLOR_ATOL               # nothing masked
                rel = np.abs(hip["depth"][0] - ref["depth"][0]) / np.maximum(1.0, np.abs(ref["depth"][0]))
                assert rel.max() <= hp.DEPTH_RTOL
                hp.compare_grads(hip["grads"], ref["grads"], names=names)                      # no row exempt
            else:
                hp.compare_forward(hip, ref)
                hp.compare_grads_by_row(hip, ref, P)
            for k in names:
                b = ref["grads"][k].reshape(P, -1).astype(np.float64)
                total[k] = b if k not in total else total[k] + b
            fy, fx = np.nonzero(st["fragile"] != 0)
            flagged.append((st["conic_opacity"].copy(), st["means2D"].copy(), fx.astype(np.float64), fy.astype(np.float64)))
            del ref, hip, st

        # the step: one C call, three views in flight, gradients accumulated into zeroed tensors
        dev = hip_device
        d = {k: v.to(dev).contiguous() for k, v in cloud.items()}
        cams = [path[i].to(dev) for i in view_ids]
        batch = parallel.ViewBatch(cams, [g.to(dev)] * 3, 3, bg.to(dev), int(worst_R * 1.25) + 4096, n_streams=3)
        acc = {"means3D": torch.zeros(P, 3, device=dev), "means2D": torch.zeros(P, 3, device=dev),
               "opacity": torch.zeros(P, 1, device=dev), "sh": torch.zeros(P, 16, 3, device=dev),
               "scales": torch.zeros(P, 3, device=dev), "rotations": torch.zeros(P, 4, device=dev)}
        batch.run(d["means3D"], d["opacities"], d["scales"], d["rotations"], d["shs"], acc)
        batch.check()
        assert _lib.last_launch_shapes() == ("tile", "tile")
        report = {}
        for k in names:
            a = acc[k].cpu().numpy().reshape(P, -1).astype(np.float64)
            b = total[k]
            scale = float(np.abs(b).max())
            row = np.abs(a - b).max(axis=1)
            bad = np.nonzero(row > hp.GRAD_RTOL * scale)[0]
            report[k] = (f"{row.max() / scale:.2e}", len(bad))
            if strict:
                assert len(bad) == 0, (k, report[k])
                continue
            assert len(bad) <= 32 and row.max() <= 1.5e-3 * scale, (k, report[k])
            for i in bad:
                touched = False
                for co, m2, fx, fy in flagged:
                    ca, cb, cc, op = co[i].astype(np.float64)
                    dx, dy = m2[i, 0] - fx, m2[i, 1] - fy
                    power = -0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy
                    touched |= bool(((power <= 1e-6) & (op * np.exp(np.minimum(power, 0.0)) >= 0.9 / 255.0)).any())
                assert touched, f"{k}: Gaussian {i} differs by {row[i] / scale:.2e} and touches no threshold-fragile pixel of any view"
        print("C3 headline step", "strict" if strict else "default", report)
    finally:
        config.set_strict_parity(False)
        config.reset()


def test_backward_with_a_smaller_bound_than_its_forward_is_reported(hip_device):
    """lr_backward sizes the blend backward's launch from R / binning_capacity (one workgroup per listed 256-position segment).
    The one-wave-per-tile kernel takes exactly one segment per workgroup, so a caller that passes a SMALLER R than its forward
    returned leaves segments without a workgroup: the kernel notices (GeomHeader::bwd_uncovered) and lr_check on the view's geom
    buffer -- and lr_backward itself under debug -- report it instead of returning incomplete gradients silently."""
    from luciddreamer_amd import _C, _lib
    P, W, H = 150_000, 160, 96                      # thousands of instances per tile: dozens of listed segments
    cam, cloud = hp.box_setup(P, W, H)
    dev = hip_device
    bg = torch.zeros(3, device=dev)
    out = _raw_forward(cloud, cam, 3, bg.cpu(), dev)
    R, color, depth, radii, geom, binning, img = out
    assert R > 60 * 256 * 4
    tfx, tfy = hp.tan_fov(cam)
    c = cam.to(dev)
    e = torch.Tensor([])
    d = {k: v.to(dev) for k, v in cloud.items()}
    g = synthetic.upstream_grad(H, W).to(dev)
    gd = torch.zeros(1, H, W, device=dev)

    def backward(r, debug):
        return _C.rasterize_gaussians_backward(bg, d["means3D"], radii, e, d["scales"], d["rotations"], 1.0, e,
                                               c.world_view_transform, c.full_proj_transform, tfx, tfy, g, gd, d["shs"], 3,
                                               c.camera_center, geom, r, binning, img, debug)
    _lib.tune_set("blend_quad", 2)
    try:
        full = backward(R, False)
        torch.cuda.synchronize()
        _C.check(geom)                               # covered: nothing to report
        backward(256, False)                         # a bound that covers 3 listed segments
        with pytest.raises(RuntimeError, match="smaller than the forward"):
            _C.check(geom)
        with pytest.raises(RuntimeError, match="smaller than the forward"):
            backward(256, True)
        again = backward(R, True)                    # the forward's own R: complete again (the flag describes the LAST backward)
        for a, b in zip(full, again):
            if a is not None:
                assert torch.equal(a, b)
    finally:
        _lib.tune_set("blend_quad", -1)


def test_c4_shape_1440p_with_depth(hip_device):
    """BASELINE.json configs[3] shape at a size the oracle finishes quickly: 1440p, depth branch checked."""
    cam, cloud = hp.box_setup(300_000, 2560, 1440)
    bg = torch.zeros(3)
    ref = hp.run_oracle(cloud, cam, 3, bg)
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device)
    fig = hp.compare_forward(hip, ref, max_fragile=5e-4 * 2560 * 1440)      # flagged by the oracle: 2.1e-4 of the pixels at this overdraw
    assert (ref["depth"] > 0).mean() > 0.1
    print("C4-shape", ref["num_rendered"], fig)


def test_c3_full_size_properties(hip_device):
    """1M Gaussians, 1080p (BASELINE.json configs[2] shape): properties that need no oracle run."""
    P, W, H = 1_000_000, 1920, 1080
    cloud = synthetic.make_cloud(P, "band", 0)
    cam = cameras.rotate360_path(W, H, n_views=30)[7]
    bg = torch.zeros(3)
    out1 = _raw_forward(cloud, cam, 3, bg, hip_device)
    out2 = _raw_forward(cloud, cam, 3, bg, hip_device)
    assert out1[0] > 100_000
    # forward is deterministic (no atomics on the forward path)
    assert torch.equal(out1[1], out2[1]) and torch.equal(out1[2], out2[2]) and torch.equal(out1[3], out2[3])
    u = _unpack(out1, P, W, H)
    R = u["point_list"].shape[0]                                        # instances after exact tile culling
    assert 0 < R <= out1[0]
    rng = u["ranges"].astype(np.int64)
    assert int((rng[:, 1] - rng[:, 0]).sum()) == R                      # ranges partition the instance list
    radii = out1[3].cpu().numpy()
    assert set(np.unique(u["point_list"])).issubset(set(np.nonzero(radii > 0)[0]))
    depth = u["rec"][:, 9]
    nonempty = np.nonzero(rng[:, 1] > rng[:, 0])[0]
    for t in nonempty[:: max(1, len(nonempty) // 200)]:
        ids = u["point_list"][rng[t, 0]:rng[t, 1]]
        d = depth[ids]
        assert np.all(d[1:] >= d[:-1])                                   # sortedness per tile
        same = d[1:] == d[:-1]
        assert np.all(ids[1:][same] > ids[:-1][same])
    assert np.all(u["n_contrib"] <= (rng[:, 1] - rng[:, 0]).max())
    T = u["final_T"]
    assert T.min() >= 0.0 and T.max() <= 1.0
    # async mode with a generous capacity reproduces the exact-mode image bit for bit
    out3 = _raw_forward(cloud, cam, 3, bg, hip_device, binning_capacity=int(R * 1.3) + 4096)
    assert out3[0] == -1 and torch.equal(out3[1], out1[1]) and torch.equal(out3[2], out1[2])

    # backward: linear in the upstream gradient, zero for culled Gaussians
    g = synthetic.upstream_grad(H, W)
    h1 = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    h2 = hp.run_hip(cloud, cam, 3, bg, hip_device, 2.0 * g)
    for k in ("means3D", "sh", "scales", "rotations", "opacity", "means2D"):
        a, b = h1["grads"][k], h2["grads"][k]
        assert np.abs(2.0 * a - b).max() <= 2e-4 * np.abs(b).max(), k     # float atomics reorder sums run to run
        assert np.abs(a[radii <= 0]).max() == 0.0, k


def test_knn_parity_and_properties(hip_device):
    from oracle import oracle
    from simple_knn._C import distCUDA2
    pts = synthetic.make_cloud(20_000, "box", 4)["means3D"]
    got = distCUDA2(pts.to(hip_device)).cpu().numpy()
    ref = oracle.dist2(pts.numpy())
    assert np.array_equal(got, ref) or np.abs(got - ref).max() <= 1e-6 * ref.max()
    # tiny and degenerate inputs
    for P in (1, 2, 3, 4, 257):
        p = torch.rand(P, 3, generator=torch.Generator().manual_seed(P))
        a = distCUDA2(p.to(hip_device)).cpu().numpy()
        b = oracle.dist2(p.numpy())
        assert np.allclose(a, b, rtol=1e-6), P
    # large: permutation equivariance (result is written at the original index)
    big = synthetic.make_cloud(300_000, "band", 5)["means3D"]
    perm = torch.randperm(300_000, generator=torch.Generator().manual_seed(0))
    d1 = distCUDA2(big.to(hip_device)).cpu()
    d2 = distCUDA2(big[perm].contiguous().to(hip_device)).cpu()
    assert torch.equal(d1[perm], d2)
    assert float(d1.min()) > 0
