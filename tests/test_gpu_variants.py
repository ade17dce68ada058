"""GPU: the kernel variants behind lr_tune_set are interchangeable -- the pooled and the thread-per-Gaussian preprocess
kernels produce IDENTICAL forward outputs (bit for bit: same arithmetic, different work distribution), the two reductions
of the blend backward the same gradients up to float summation order.  Each variant is also put through the oracle
comparison of tests/helpers.py."""
import numpy as np
import pytest
import torch

from luciddreamer_amd import _lib, cameras, synthetic
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _needs_diagnostics():
    """Retired kernels (the round-2 LDS-atomic reduction, the scanned partition) are compiled into the diagnostics build only
    (`python -m luciddreamer_amd.build --diagnostics`, tools/diag_env.sh python -m pytest ...): the product library ships none."""
    if not _lib.diagnostics_build():
        pytest.skip("retired kernel: diagnostics build only (tools/diag_env.sh)")


@pytest.fixture(autouse=True)
def _restore_knobs():
    yield
    for k in ("preprocess", "bwd_red", "hit_mask", "walk_own", "tsort", "part_scan", "bwd_seg", "fwd_pair", "blend_quad",
              "views_in_flight", "strict"):
        _lib.tune_set(k, -1)


def _run(cloud, cam, dev, g):
    return hp.run_hip(cloud, cam, 3, torch.zeros(3), dev, g)


@pytest.mark.parametrize("kind,P,W,H", [("band", 60_000, 640, 360), ("box", 30_000, 480, 272), ("band", 700, 200, 120)])
def test_preprocess_kernels_are_bit_identical(hip_device, kind, P, W, H):
    cloud = synthetic.make_cloud(P, kind, 3)
    cam = cameras.rotate360_path(W, H, n_views=30)[7] if kind == "band" else cameras.identity_camera(W, H)
    g = synthetic.upstream_grad(H, W)
    outs = []
    for v in (0, 1):
        _lib.tune_set("preprocess", v)
        outs.append(_run(cloud, cam, hip_device, g))
    a, b = outs
    assert np.array_equal(a["radii"], b["radii"])
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"])
    for k in a["grads"]:
        assert np.array_equal(a["grads"][k], b["grads"][k]), k
    ref = hp.run_oracle(cloud, cam, 3, torch.zeros(3), g)
    hp.compare_forward(b, ref)                                 # the pooled kernel against the oracle (radii exact, 1e-5)
    hp.compare_grads(b["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_pooled_preprocess_raw_mode_and_precomputed_inputs(hip_device):
    """colors_precomp / cov3D_precomp take the other branches of the pooled kernel's two halves."""
    cam, cloud = hp.box_setup(5000, 256, 160)
    cols = torch.rand(5000, 3, generator=torch.Generator().manual_seed(5))
    g = synthetic.upstream_grad(160, 256)
    outs = []
    for v in (0, 1):
        _lib.tune_set("preprocess", v)
        outs.append(hp.run_hip(cloud, cam, 3, torch.zeros(3), hip_device, g, colors_precomp=cols))
    assert np.array_equal(outs[0]["color"], outs[1]["color"]) and np.array_equal(outs[0]["radii"], outs[1]["radii"])
    for k in outs[0]["grads"]:
        assert np.array_equal(outs[0]["grads"][k], outs[1]["grads"][k]), k


@pytest.mark.parametrize("W,H", [(640, 360), (1280, 720)])      # QUAD shape / two-wave shape of k_render_bwd
def test_backward_reductions_agree(hip_device, W, H):
    _needs_diagnostics()
    cam, cloud = hp.box_setup(40_000, W, H)
    g = synthetic.upstream_grad(H, W)
    outs = []
    for v in (0, 1):
        _lib.tune_set("bwd_red", v)
        outs.append(_run(cloud, cam, hip_device, g))
    for k in outs[0]["grads"]:
        a, b = outs[0]["grads"][k], outs[1]["grads"][k]
        scale = float(np.abs(a).max())
        assert float(np.abs(a - b).max()) <= 4e-6 * scale, k
    # both repeatable bit for bit
    _lib.tune_set("bwd_red", 1)
    again = _run(cloud, cam, hip_device, g)
    for k in again["grads"]:
        assert np.array_equal(again["grads"][k], outs[1]["grads"][k]), k


@pytest.mark.parametrize("shape", [0, 1, 2])        # 2 waves x 2 pixels, 4 waves x 1 pixel, 1 wave x 4 pixels per tile
@pytest.mark.parametrize("scale_mult", [1.0, 4.0])  # sparse overdraw (hardly a pixel stops early) / heavy overdraw (most do)
def test_check_free_candidate_loop_is_bit_identical(hip_device, shape, scale_mult):
    """A wave none of whose pixels the forward stopped early walks a copy of the candidate loop without the `pos < last` test
    (render_bwd.hip bwd_pixel CHECK_LAST); lr_tune_set("bwd_red", 4) sends every wave through the copy WITH it.  The test can
    only fail for stopped pixels, so the two must give the same bits -- on a scene where nearly every wave takes the shortcut
    and on one where nearly none does."""
    cam, cloud = hp.box_setup(30_000, 640, 360, seed=3, scale_mult=scale_mult)
    g = synthetic.upstream_grad(360, 640)
    _lib.tune_set("blend_quad", shape)
    try:
        _lib.tune_set("bwd_red", 4)
        a = _run(cloud, cam, hip_device, g)
        _lib.tune_set("bwd_red", -1)
        b = _run(cloud, cam, hip_device, g)
    finally:
        _lib.tune_set("bwd_red", -1)
        _lib.tune_set("blend_quad", -1)
    for k in a["grads"]:
        assert np.array_equal(a["grads"][k], b["grads"][k]), k
    assert any(np.abs(a["grads"][k]).max() > 0 for k in a["grads"])


def test_backward_shapes_differentiate_the_layers_the_forward_blended(hip_device):
    """The backward re-decides `alpha >= 1/255` per (pixel, candidate) pair, so its alpha must be the forward's BIT FOR BIT
    (common.h gauss_bd / gauss_cdd / gauss_power1: explicit roundings and FMAs).  Before round 4 the compiler contracted the
    shared source differently per call site: the forward and the one-pixel-per-lane shape fused the last product of the
    exponent into the final add, the two- and four-pixel shapes rounded it first -- on C3 views 15 and 18 one to four
    Gaussians per view sat on a pixel where that ulp decided the 1/255 test, their gradient rows moved by up to 1.5e-3 of
    the tensor's maximum and the shapes disagreed with each other (tools/shape_vs_oracle.py).  Now every shape must give
    the same rows to float summation order."""
    P = 1_000_000
    cloud = synthetic.make_cloud(P, "band", 0)
    path = cameras.rotate360_path(1920, 1080, n_views=30)
    g = synthetic.upstream_grad(1080, 1920)
    try:
        for vi in (15, 18):
            outs = []
            for shape in (0, 1, 2):
                _lib.tune_set("blend_quad", shape)
                outs.append(hp.run_hip(cloud, path[vi], 3, torch.zeros(3), hip_device, g)["grads"])
            for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"):
                scale = float(np.abs(outs[0][k]).max())
                for o in outs[1:]:
                    assert float(np.abs(o[k] - outs[0][k]).max()) <= 4e-6 * scale, (vi, k)
    finally:
        _lib.tune_set("blend_quad", -1)


@pytest.mark.parametrize("W,H", [(640, 360), (1280, 720)])      # 4-wave / 2-wave shape of k_render_bwd
def test_strict_mode_against_the_oracle_without_exemptions(hip_device, W, H):
    """Strict evaluation (config.set_strict_parity): image, depth and gradients against the CPU oracle with NO pixel masked
    and no row exempt -- the oracle evaluates the same float operations."""
    from luciddreamer_amd import config
    cam, cloud = hp.box_setup(40_000, W, H, seed=11, scale_mult=1.5)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.1, 0.0, 0.2])
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    config.set_strict_parity(True)
    try:
        hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    finally:
        config.set_strict_parity(False)
    assert np.array_equal(hip["radii"], ref["radii"])
    cerr = np.abs(hip["color"] - ref["color"]).max()
    derr = (np.abs(hip["depth"][0] - ref["depth"][0]) / np.maximum(1.0, np.abs(ref["depth"][0]))).max()
    assert cerr <= hp.COLOR_ATOL and derr <= hp.DEPTH_RTOL, (float(cerr), float(derr))
    hp.compare_grads(hip["grads"], ref["grads"], names=("means2D", "opacity", "means3D", "sh", "scales", "rotations"))


@pytest.mark.parametrize("fused", [False, True])
def test_compiled_and_python_autograd_nodes_agree(hip_device, fused):
    """The operator's autograd node exists twice: compiled (csrc/torch_ext.cpp RasterizeFn, the default) and in Python
    (rasterizer._RasterizeGaussians, used with settings.debug because it writes the reference's snapshot dumps).  Same C
    calls underneath: images and gradients must be bit-identical, also with fused gradient accumulation into existing
    .grad tensors and with an upstream gradient on the depth output."""
    from luciddreamer_amd import config
    cloud = synthetic.make_cloud(20_000, "band", 11)
    cam = cameras.rotate360_path(400, 240, n_views=30)[4]
    g, gd = synthetic.upstream_grad(240, 400), torch.rand(1, 240, 400)
    config.set_fused_grad_accumulation(fused)
    try:
        a = hp.run_hip(cloud, cam, 3, torch.zeros(3), hip_device, g, debug=False, grad_depth=gd)
        b = hp.run_hip(cloud, cam, 3, torch.zeros(3), hip_device, g, debug=True, grad_depth=gd)
    finally:
        config.set_fused_grad_accumulation(False)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["radii"], b["radii"])
    assert "RasterizeFn" in a["color_t"].grad_fn.name() and "RasterizeGaussians" in b["color_t"].grad_fn.name()
    for k in a["grads"]:
        assert np.array_equal(a["grads"][k], b["grads"][k]), k
    assert float(np.abs(a["grads"]["means3D"]).max()) > 0


def test_compiled_node_accumulates_into_existing_grads_and_checks_versions(hip_device):
    """Two views through the compiled node with fused accumulation: .grad ends up the sum of the two views' gradients (first
    view: no .grad yet -> autograd installs the dense tensor; second: `+=` inside the kernel).  And an in-place change of a
    saved input between forward and backward is an error, as with the reference's ctx.save_for_backward."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import config
    dev = hip_device
    cloud = synthetic.make_cloud(8_000, "band", 2)
    cams = [c.to(dev) for c in cameras.rotate360_path(320, 200, n_views=30)[3:5]]
    g = synthetic.upstream_grad(200, 320).to(dev)

    def leafs():
        return {k: v.detach().to(dev).requires_grad_(True) for k, v in cloud.items()}

    def render(p, cam, m2d):
        tfx, tfy = hp.tan_fov(cam)
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, tfx, tfy, torch.zeros(3, device=dev), 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        return GaussianRasterizer(rs)(means3D=p["means3D"], means2D=m2d, opacities=p["opacities"], shs=p["shs"],
                                      scales=p["scales"], rotations=p["rotations"])[0]
    sums = []
    for fused in (False, True):
        config.set_fused_grad_accumulation(fused)
        try:
            p = leafs()
            for cam in cams:
                m2d = torch.zeros_like(p["means3D"], requires_grad=True)
                (render(p, cam, m2d) * g).sum().backward()
        finally:
            config.set_fused_grad_accumulation(False)
        sums.append({k: v.grad.cpu().numpy() for k, v in p.items()})
    for k in sums[0]:
        scale = float(np.abs(sums[0][k]).max())
        assert float(np.abs(sums[0][k] - sums[1][k]).max()) <= 2e-6 * scale, k       # same two addends per element
    p = leafs()
    m2d = torch.zeros_like(p["means3D"], requires_grad=True)
    color = render(p, cams[0], m2d)
    with torch.no_grad():
        p["scales"].mul_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        (color * g).sum().backward()


@pytest.mark.parametrize("kind,P,W,H,scale_mult", [("band", 60_000, 640, 360, 1.0), ("box", 4_000, 800, 450, 6.0)])
def test_binning_walks_with_and_without_hit_masks_agree(hip_device, kind, P, W, H, scale_mult):
    """The binning walks a Gaussian's instances from the bit mask preprocess leaves (common.h HitRec; rectangles of up to 64
    tiles whose origin fits the record) or, without one, from its record, repeating the tile test (rectangles of 65-96 tiles,
    everything on tile grids beyond 4096).  lr_tune_set("hit_mask", 0) takes the masks away from every Gaussian: same lists,
    same images, same gradients, bit for bit.  The second scene has splats that cover hundreds of tiles (unculled walk)."""
    cloud = synthetic.make_cloud(P, kind, 5, scale_mult=scale_mult)
    cam = cameras.rotate360_path(W, H, n_views=30)[9] if kind == "band" else cameras.identity_camera(W, H)
    g = synthetic.upstream_grad(H, W)
    outs = []
    try:
        for v in (-1, 0):
            _lib.tune_set("hit_mask", v)
            outs.append(_run(cloud, cam, hip_device, g))
    finally:
        _lib.tune_set("hit_mask", -1)
    a, b = outs
    assert np.array_equal(a["radii"], b["radii"])
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"])
    for k in a["grads"]:
        assert np.array_equal(a["grads"][k], b["grads"][k]), k
    ref = hp.run_oracle(cloud, cam, 3, torch.zeros(3), g)
    hp.compare_forward(a, ref, max_fragile=max(8, 1e-3 * W * H))      # huge splats: heavy overdraw, many pixels near a threshold


@pytest.mark.parametrize("W,H", [(640, 360), (1280, 720)])      # quadrant-per-wave shape / two-wave shape of k_render_bwd
@pytest.mark.parametrize("red", [0, 1])
def test_backward_is_bit_repeatable_over_many_runs(hip_device, W, H, red):
    """Forty runs of the same view give the same gradients, bit for bit, with either reduction.  This is the test that would
    have caught the missing `s_waitcnt lgkmcnt(0)` in front of the barrier that closes a batch of k_render_bwd (common.h
    lds_barrier): with the LDS-atomic reduction one Gaussian's colour gradient differed in ~5 % of the runs."""
    if red == 0:
        _needs_diagnostics()
    cam, cloud = hp.box_setup(40_000, W, H)
    g = synthetic.upstream_grad(H, W)
    _lib.tune_set("bwd_red", red if red == 0 else -1)
    first = _run(cloud, cam, hip_device, g)
    for _ in range(40):
        again = _run(cloud, cam, hip_device, g)
        for k in first["grads"]:
            assert np.array_equal(again["grads"][k], first["grads"][k]), k


def test_mid_size_bins_bucket_sort_and_network_give_the_same_lists(hip_device):
    """Bins of 257..1024 entries are put in order by a bucket sort (tilebin.hip bucket_sort_bin in k_tile_sort_small) or,
    with lr_tune_set("tsort", 0), by the bitonic network: the sort word is a total order, so lists, images and gradients are
    bit-identical.  Dense box cloud at 480 x 272: ~500 instances per tile."""
    cam, cloud = hp.box_setup(120_000, 480, 272, seed=9)
    g = synthetic.upstream_grad(272, 480)
    outs = []
    try:
        for v in (-1, 1, 0):          # default (bucket sort from 65 entries), bucket sort from 257, network only
            _lib.tune_set("tsort", v)
            outs.append(_run(cloud, cam, hip_device, g))
    finally:
        _lib.tune_set("tsort", -1)
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"])
        for k in a["grads"]:
            assert np.array_equal(a["grads"][k], b["grads"][k]), k
    ref = hp.run_oracle(cloud, cam, 3, torch.zeros(3))
    hp.compare_forward(a, ref, max_fragile=max(8, 1e-3 * 480 * 272))


@pytest.mark.parametrize("kind,P,W,H,scale_mult", [("band", 60_000, 640, 360, 1.0), ("box", 30_000, 480, 272, 3.0),
                                                   ("box", 200_000, 512, 512, 1.0), ("box", 4000, 2560, 1440, 6.0)])
def test_ranges_reserved_by_the_count_kernel_give_the_lists_of_the_scanned_rows(hip_device, kind, P, W, H, scale_mult):
    """Default: every workgroup of the count kernel reserves its range inside a bin with one returning atomic on the bin's
    cursor -- which workgroup gets which range depends on arrival; lr_tune_set("part_scan", 1): per-workgroup counts and a scan
    kernel (rounds 2-4), ranges in workgroup order.  The per-bin sort is a total order on the words, so lists, images and
    gradients must be the same bits either way -- also run to run (several passes: arrival order differs)."""
    _needs_diagnostics()
    cloud = synthetic.make_cloud(P, kind, 11)
    if scale_mult != 1.0:
        cloud["scales"] = cloud["scales"] * scale_mult
    cam = cameras.rotate360_path(W, H, n_views=30)[3] if kind == "band" else cameras.identity_camera(W, H)
    g = synthetic.upstream_grad(H, W)
    outs = []
    for v in (1, -1, -1, -1):
        _lib.tune_set("part_scan", v)
        outs.append(_run(cloud, cam, hip_device, g))
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["radii"], b["radii"])
        assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"])
        for k in a["grads"]:
            assert np.array_equal(a["grads"][k], b["grads"][k]), k


@pytest.mark.parametrize("shape", [0, 1, 2])        # 2 waves x 2 pixels, 4 waves x 1 pixel, 1 wave x 4 pixels per tile
@pytest.mark.parametrize("P,W,H,scale_mult", [(150_000, 256, 192, 1.0), (40_000, 320, 208, 5.0)])
def test_list_segments_give_the_gradients_of_the_whole_list(hip_device, shape, P, W, H, scale_mult):
    """Lists longer than 256 instances are differentiated in segments, one workgroup each, starting from the state the forward
    left at every 256th position (common.h BWD_SEG; render_fwd.hip checkpoints): T there is the forward's own, the colour behind
    it the final colour minus the colour so far.  lr_tune_set("bwd_seg", 0) = one workgroup walks the whole list (rounds 1-4).
    Same layers differentiated, same per-instance slots written: the gradients agree to float rounding, both agree with the
    oracle, and both are repeatable bit for bit.  Dense clouds on small images: hundreds to thousands of instances per tile,
    most pixels stopped early in the second scene."""
    cam, cloud = hp.box_setup(P, W, H, seed=13, scale_mult=scale_mult)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.3, 0.1, 0.2])
    _lib.tune_set("blend_quad", shape)
    try:
        outs = []
        for seg in (0, 1, 1):
            _lib.tune_set("bwd_seg", seg)
            outs.append(hp.run_hip(cloud, cam, 3, bg, hip_device, g))
    finally:
        _lib.tune_set("bwd_seg", -1)
        _lib.tune_set("blend_quad", -1)
    whole, segs, again = outs
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    R = int(ref["num_rendered"])
    assert R > 256 * 3 * ((W + 15) // 16) * ((H + 15) // 16)          # several segments per tile on average (before exact culling)
    assert np.array_equal(whole["color"], segs["color"])
    for k in whole["grads"]:
        a, b = whole["grads"][k], segs["grads"][k]
        scale = float(np.abs(a).max())
        assert float(np.abs(a - b).max()) <= 4e-6 * scale, (k, float(np.abs(a - b).max()) / max(scale, 1e-30))
        assert np.array_equal(segs["grads"][k], again["grads"][k]), k
    assert all(float(np.abs(whole["grads"][k]).max()) > 0 for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"))
    hp.compare_grads_by_row(segs, ref, P)


@pytest.mark.parametrize("P,W,H,scale_mult", [(60_000, 640, 360, 1.0), (40_000, 320, 208, 5.0), (20_000, 1280, 720, 2.0)])
def test_forward_candidate_pairs_give_the_same_bits(hip_device, P, W, H, scale_mult):
    """Large images with other views in flight take the one-wave-per-tile kernel (a lane owns four pixels, a candidate's fields are
    read once per tile); lr_tune_set("fwd_pair", 0 / 2) forces quadrant / tile (1: round 5's candidate-pair variant of the quadrant
    kernel -- both alphas side by side, then the two steps of the T recursion in order -- retired to the diagnostics build).
    Same operations per pixel and candidate in the same order: images, depth, n_contrib (through the gradients) and checkpoints
    are the same bits."""
    cam, cloud = hp.box_setup(P, W, H, seed=21, scale_mult=scale_mult)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.3, 0.1, 0.2])
    outs = []
    variants = (0, 1, 2) if _lib.diagnostics_build() else (0, 2)
    try:
        for v in variants:
            _lib.tune_set("fwd_pair", v)
            outs.append(hp.run_hip(cloud, cam, 3, bg, hip_device, g))
    finally:
        _lib.tune_set("fwd_pair", -1)
    a = outs[0]
    for v, b in zip(variants[1:], outs[1:]):
        assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["depth"], b["depth"]), v
        for k in a["grads"]:
            assert np.array_equal(a["grads"][k], b["grads"][k]), (v, k)


def _stack_cloud(n, W, H, cam, px, py, opacity, seed):
    """n small Gaussians stacked behind pixel (px, py) at increasing depths (jittered by a pixel or two): one tile whose list is
    exactly n long -- the segment boundaries of the blend backward (common.h BWD_SEG = 256) at chosen positions."""
    import math
    g = torch.Generator().manual_seed(seed)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    z = torch.linspace(2.0, 6.0, n)
    jx = px + (torch.rand(n, generator=g) - 0.5) * 3.0
    jy = py + (torch.rand(n, generator=g) - 0.5) * 3.0
    x = ((jx + 0.5) * 2.0 / W - 1.0) * tfx * z
    y = ((jy + 0.5) * 2.0 / H - 1.0) * tfy * z
    focal = W / (2.0 * tfx)
    sig = (1.0 + torch.rand(n, generator=g)) * z / focal                     # 1-2 pixels of standard deviation on the screen
    scales = torch.stack([sig, sig * (0.6 + 0.8 * torch.rand(n, generator=g)), sig], dim=1)
    q = torch.randn(n, 4, generator=g)
    shs = torch.zeros(n, 16, 3)
    shs[:, 0, :] = (torch.rand(n, 3, generator=g) - 0.5) / 0.28209479177387814
    shs[:, 1:, :] = 0.05 * torch.randn(n, 15, 3, generator=g)
    return dict(means3D=torch.stack([x, y, z], dim=1).float().contiguous(), scales=scales.float().contiguous(),
                rotations=(q / q.norm(dim=1, keepdim=True)).float().contiguous(),
                opacities=torch.full((n, 1), float(opacity)), shs=shs.float().contiguous())


@pytest.mark.parametrize("shape", [0, 1, 2])
@pytest.mark.parametrize("opacity", [0.006, 0.08])      # nobody stops early / the centre pixels stop inside the first segment
@pytest.mark.parametrize("n", [255, 256, 257, 511, 512, 513, 700])
def test_list_segments_at_their_boundaries(hip_device, shape, opacity, n):
    """A tile whose list is exactly n long for n around the multiples of the segment length: no segment (256), a segment of one
    element (257, 513), whole segments (512), a ragged last one (700) -- against the whole-list walk and against the oracle, with
    pixels that never stop and with pixels that stop in front of the first checkpoint."""
    W, H = 64, 48
    cam = cameras.identity_camera(W, H)
    cloud = _stack_cloud(n, W, H, cam, 40.0, 24.0, opacity, seed=n)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.2, 0.4, 0.1])
    _lib.tune_set("blend_quad", shape)
    try:
        outs = []
        for seg in (0, 1):
            _lib.tune_set("bwd_seg", seg)
            outs.append(hp.run_hip(cloud, cam, 3, bg, hip_device, g))
    finally:
        _lib.tune_set("bwd_seg", -1)
        _lib.tune_set("blend_quad", -1)
    whole, segs = outs
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hp.compare_forward(segs, ref)
    for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"):
        a, b = whole["grads"][k], segs["grads"][k]
        scale = float(np.abs(a).max())
        assert scale > 0 and float(np.abs(a - b).max()) <= 4e-6 * scale, (k, float(np.abs(a - b).max()) / scale)
    hp.compare_grads(segs["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


@pytest.mark.parametrize("strict", [0, 1])
@pytest.mark.parametrize("n,opacity,W,H", [(257, 0.006, 64, 48), (513, 0.08, 64, 48), (700, 0.006, 70, 41), (1100, 0.02, 70, 41)])
def test_tile_forward_with_segments_and_ragged_image(hip_device, strict, n, opacity, W, H):
    """The one-wave-per-tile forward (render_fwd.hip k_render_fwd_tile: 64 list entries per round, a checkpoint every 256) on
    lists that cross segment boundaries, on an image whose last tile column and row are cut (70 x 41: quadrants partly and wholly
    outside), in both arithmetic modes: the quadrant kernel's bits -- image, depth, every gradient through every backward shape --
    and the oracle's values."""
    cam = cameras.identity_camera(W, H)
    cloud = _stack_cloud(n, W, H, cam, W - 9.0, H - 5.0, opacity, seed=n + 1)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.2, 0.4, 0.1])
    _lib.tune_set("strict", strict)
    try:
        for shape in (0, 1, 2):
            _lib.tune_set("blend_quad", shape)
            outs = []
            for v in (0, 2):                        # quadrant kernel, one wave per tile
                _lib.tune_set("fwd_pair", v)
                outs.append(hp.run_hip(cloud, cam, 3, bg, hip_device, g))
            a, b = outs[0], outs[1]
            for o in outs[1:]:
                assert np.array_equal(a["color"], o["color"]) and np.array_equal(a["depth"], o["depth"]), shape
                for k in a["grads"]:
                    assert np.array_equal(a["grads"][k], o["grads"][k]), (shape, k)
    finally:
        _lib.tune_set("fwd_pair", -1)
        _lib.tune_set("blend_quad", -1)
        _lib.tune_set("strict", -1)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hp.compare_forward(b, ref)
    hp.compare_grads(b["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_pixel_sized_gaussians_in_one_layer_against_the_oracle(hip_device):
    """LucidDreamer's own scene statistics (synthetic.make_cloud kind "shell": one layer of pixel-sized isotropic Gaussians lifted
    from a panorama, bench.py workload ld512) at a tenth of the size: lists of ~300 per tile of splats a few pixels wide --
    image, depth, radii and gradients against the oracle; segments on and off agree."""
    cloud = synthetic.make_cloud(100_000, "shell", 0)
    W = H = 162
    cam = cameras.rotate360_path(W, H, n_views=30)[3]
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.1, 0.1, 0.1])
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    outs = []
    try:
        for seg in (1, 0):
            _lib.tune_set("bwd_seg", seg)
            outs.append(hp.run_hip(cloud, cam, 3, bg, hip_device, g))
    finally:
        _lib.tune_set("bwd_seg", -1)
    a, b = outs
    assert int((ref["radii"] > 0).sum()) > 10_000
    hp.compare_forward(a, ref, max_fragile=max(8, 1e-3 * W * H))
    hp.compare_grads_by_row(a, ref, 100_000)
    for k in a["grads"]:
        scale = float(np.abs(b["grads"][k]).max())
        assert float(np.abs(a["grads"][k] - b["grads"][k]).max()) <= 4e-6 * scale, k
