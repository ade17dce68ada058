"""GPU parity tests: the HIP rasterizer (public Python API -> C-ABI) against the CPU oracle on
identical seeded inputs.  Tolerances are the ones stated in tests/helpers.py."""
import numpy as np
import pytest
import torch

from luciddreamer_amd import cameras, synthetic
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_forward_backward_small(hip_device, degree):
    """C1-shaped case (10k Gaussians, 256x256) for every SH degree, forward + backward."""
    cam, cloud = hp.box_setup(10_000, 256, 256)
    bg = torch.tensor([0.1, 0.2, 0.3])
    g = synthetic.upstream_grad(256, 256)
    ref = hp.run_oracle(cloud, cam, degree, bg, g)
    hip = hp.run_hip(cloud, cam, degree, bg, hip_device, g)
    fig = hp.compare_forward(hip, ref)
    gfig = hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])
    print(fig, gfig)


def test_sh_layout_M1(hip_device):
    """sh of shape (P,1,3) with degree 0 ('pass sh (P,1,3) ... with degree 0')."""
    cam, cloud = hp.box_setup(5_000, 200, 120, sh_coeffs=1)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(120, 200)
    ref = hp.run_oracle(cloud, cam, 0, bg, g)
    hip = hp.run_hip(cloud, cam, 0, bg, hip_device, g)
    hp.compare_forward(hip, ref)
    hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_ragged_image_and_big_splats(hip_device):
    """Image size not a multiple of 16; large Gaussians exercise the wave-cooperative emission."""
    cam, cloud = hp.box_setup(3_000, 250, 131, scale_mult=6.0)
    bg = torch.tensor([1.0, 0.5, 0.0])
    g = synthetic.upstream_grad(131, 250)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    hp.compare_forward(hip, ref)
    hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_precomputed_colors_and_cov(hip_device):
    """colors_precomp + cov3D_precomp input path (reference gaussian_renderer/__init__.py:62-63, 81-82)."""
    from oracle import torch_oracle
    cam, cloud = hp.box_setup(4_000, 160, 160)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(160, 160)
    cov = torch_oracle.cov3d_from_scale_rot(cloud["scales"].double(), 1.0, cloud["rotations"].double()).float()
    cols = torch.rand(4_000, 3, generator=torch.Generator().manual_seed(5))
    ref = hp.run_oracle(cloud, cam, 0, bg, g, colors_precomp=cols, cov3D_precomp=cov)
    hip = hp.run_hip(cloud, cam, 0, bg, hip_device, g, colors_precomp=cols, cov3D_precomp=cov)
    hp.compare_forward(hip, ref)
    hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "colors", "opacity", "means3D", "cov3D"])


def test_scale_modifier(hip_device):
    cam, cloud = hp.box_setup(4_000, 160, 96)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(96, 160)
    ref = hp.run_oracle(cloud, cam, 2, bg, g, scale_modifier=1.7)
    hip = hp.run_hip(cloud, cam, 2, bg, hip_device, g, scale_modifier=1.7)
    hp.compare_forward(hip, ref)
    hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_depth_ties_keep_index_order(hip_device):
    """Exactly equal depths: the reference's stable sort keeps Gaussian-index order
    (rasterizer_impl.cu:98-108, 304-309).  Duplicate every point so ties are everywhere."""
    cam, cloud = hp.box_setup(1_500, 128, 128, scale_mult=2.0)
    dup = {k: torch.cat([v, v], dim=0).contiguous() for k, v in cloud.items()}
    # same position/depth, different colour -> the blend order is observable
    dup["shs"][1_500:, 0, :] = -dup["shs"][1_500:, 0, :]
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(128, 128)
    ref = hp.run_oracle(dup, cam, 0, bg, g)
    hip = hp.run_hip(dup, cam, 0, bg, hip_device, g)
    hp.compare_forward(hip, ref)
    hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_camera_path_views(hip_device):
    """Band cloud seen from rotate360 poses (pure rotations about the origin)."""
    cloud = synthetic.make_cloud(20_000, "band", 3)
    cams = cameras.rotate360_path(320, 180, n_views=5)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(180, 320)
    for cam in cams[1:4]:
        ref = hp.run_oracle(cloud, cam, 3, bg, g)
        hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
        hp.compare_forward(hip, ref)
        hp.compare_grads(hip["grads"], ref["grads"], names=["means2D", "opacity", "means3D", "sh", "scales", "rotations"])


def test_empty_and_all_culled(hip_device):
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    cam = cameras.identity_camera(64, 48).to(hip_device)
    tfx, tfy = hp.tan_fov(cam)
    bg = torch.tensor([0.3, 0.6, 0.9], device=hip_device)
    rs = GaussianRasterizationSettings(48, 64, tfx, tfy, bg, 1.0, cam.world_view_transform, cam.full_proj_transform,
                                       0, cam.camera_center, False, False)
    rast = GaussianRasterizer(rs)
    # P == 0: zero images (rasterize_points.cu:68-82), not background
    z3 = torch.zeros(0, 3, device=hip_device)
    color, radii, depth = rast(means3D=z3, means2D=z3, opacities=torch.zeros(0, 1, device=hip_device),
                               shs=torch.zeros(0, 16, 3, device=hip_device), scales=z3,
                               rotations=torch.zeros(0, 4, device=hip_device))
    assert color.shape == (3, 48, 64) and float(color.abs().max()) == 0.0 and radii.numel() == 0
    # everything behind the camera: background everywhere, radii 0, zero gradients
    cloud = synthetic.make_cloud(500, "box", 0)
    cloud["means3D"][:, 2] = -cloud["means3D"][:, 2]
    g = synthetic.upstream_grad(48, 64)
    hip = hp.run_hip(cloud, cameras.identity_camera(64, 48), 0, bg.cpu(), hip_device, g)
    assert (hip["radii"] == 0).all()
    assert np.allclose(hip["color"], bg.cpu().numpy()[:, None, None])
    assert float(np.abs(hip["depth"]).max()) == 0.0
    for k, v in hip["grads"].items():
        assert float(np.abs(v).max()) == 0.0, k


def test_single_gaussian_analytic(hip_device):
    """Isotropic Gaussian on the optical axis: peak colour = min(0.99, o) * rgb, radius = ceil(3*sqrt(s_px^2+0.3))
    (SURVEY 8c analytic case)."""
    W = H = 64
    cam = cameras.identity_camera(W, H)
    s, z, o = 0.05, 4.0, 0.8
    cloud = dict(means3D=torch.tensor([[0.0, 0.0, z]]), scales=torch.full((1, 3), s),
                 rotations=torch.tensor([[1.0, 0, 0, 0]]), opacities=torch.tensor([[o]]),
                 shs=torch.zeros(1, 16, 3))
    rgb = torch.tensor([[0.2, 0.5, 0.9]])
    hip = hp.run_hip(cloud, cam, 0, torch.zeros(3), hip_device, colors_precomp=rgb)
    focal = cameras.fov2focal(cam.FoVx, W)
    s_px2 = (s * focal / z) ** 2
    assert hip["radii"][0] == int(np.ceil(3.0 * np.sqrt(s_px2 + 0.3)))
    # the mean projects to pixel coordinate (W-1)/2 = 31.5: the 4 central pixels are 0.5 px away on each axis
    d2 = 0.5
    expect = o * np.exp(-0.5 * d2 / (s_px2 + 0.3)) * rgb.numpy()[0]
    assert np.allclose(hip["color"][:, 31, 31], expect, atol=2e-6)
    assert abs(hip["depth"][0, 31, 31] - z) < 1e-5


def test_prefiltered_trap_is_an_error(hip_device):
    """prefiltered=True with a culled point is a hard error (auxiliary.h:156-160)."""
    cam, cloud = hp.box_setup(200, 64, 64)
    cloud["means3D"][0, 2] = -1.0
    with pytest.raises(RuntimeError, match="prefiltered"):
        hp.run_hip(cloud, cam, 0, torch.zeros(3), hip_device, prefiltered=True)


def test_depth_gradient_is_ignored(hip_device):
    """A loss on depth yields zero parameter gradient in the reference (backward.cu:457-464, 539-554)."""
    cam, cloud = hp.box_setup(2_000, 96, 96)
    gz = torch.zeros(3, 96, 96)
    gd = torch.randn(1, 96, 96, generator=torch.Generator().manual_seed(2))
    hip = hp.run_hip(cloud, cam, 1, torch.zeros(3), hip_device, gz, grad_depth=gd)
    for k, v in hip["grads"].items():
        assert float(np.abs(v).max()) == 0.0, k


def test_mark_visible(hip_device):
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import oracle
    cloud = synthetic.make_cloud(10_000, "band", 1)
    cam = cameras.rotate360_path(64, 64, n_views=4)[1]
    camd = cam.to(hip_device)
    tfx, tfy = hp.tan_fov(cam)
    rs = GaussianRasterizationSettings(64, 64, tfx, tfy, torch.zeros(3, device=hip_device), 1.0,
                                       camd.world_view_transform, camd.full_proj_transform, 0, camd.camera_center,
                                       False, False)
    vis = GaussianRasterizer(rs).markVisible(cloud["means3D"].to(hip_device)).cpu().numpy()
    ref = oracle.mark_visible(cloud["means3D"].numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy())
    assert vis.dtype == np.bool_ and np.array_equal(vis, ref)


def test_async_mode_matches_exact(hip_device):
    """Async (no host sync) forward/backward gives the same result as exact mode; overflow is reported."""
    from luciddreamer_amd import _C, config
    cam, cloud = hp.box_setup(8_000, 192, 128)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(128, 192)
    exact = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    config.reset()
    config.set_async(True, headroom=1.5, warm_calls=1)
    try:
        first = hp.run_hip(cloud, cam, 3, bg, hip_device, g)      # measures exactly
        second = hp.run_hip(cloud, cam, 3, bg, hip_device, g)     # async, capacity from the high-water mark
        config.drain()
    finally:
        config.set_async(True)
        config.reset()
    for run in (first, second):
        assert np.array_equal(run["color"], exact["color"]) and np.array_equal(run["radii"], exact["radii"])
        assert np.array_equal(run["depth"], exact["depth"])
    # too small a capacity: flagged, never out of bounds
    dev = hip_device
    tfx, tfy = hp.tan_fov(cam)
    camd = cam.to(dev)
    out = _C.rasterize_gaussians(bg.to(dev), cloud["means3D"].to(dev), torch.Tensor([]), cloud["opacities"].to(dev),
                                 cloud["scales"].to(dev), cloud["rotations"].to(dev), 1.0, torch.Tensor([]),
                                 camd.world_view_transform, camd.full_proj_transform, tfx, tfy, 128, 192,
                                 cloud["shs"].to(dev), 3, camd.camera_center, False, False, binning_capacity=100)
    with pytest.raises(RuntimeError, match="capacity"):
        _C.check(out[4])


def test_header_tickets(hip_device):
    """lr_header_post / lr_header_poll (the non-blocking read-back behind async mode's deferred check): the words are the
    header's, a ticket is released by the poll that completes it, and a released or unknown ticket is an error."""
    from luciddreamer_amd import _C
    cam, cloud = hp.box_setup(8_000, 192, 128)
    dev = hip_device
    tfx, tfy = hp.tan_fov(cam)
    camd = cam.to(dev)
    args = (torch.zeros(3, device=dev), cloud["means3D"].to(dev), torch.Tensor([]), cloud["opacities"].to(dev),
            cloud["scales"].to(dev), cloud["rotations"].to(dev), 1.0, torch.Tensor([]), camd.world_view_transform,
            camd.full_proj_transform, tfx, tfy, 128, 192, cloud["shs"].to(dev), 3, camd.camera_center, False, False)
    exact = _C.rasterize_gaussians(*args)
    out = _C.rasterize_gaussians(*args, binning_capacity=10_000_000)
    tickets = [_C.header_post(out[4]) for _ in range(3)]
    assert len(set(tickets)) == 3
    words = [_C.header_poll(t, True) for t in tickets]
    assert words[0] == words[1] == words[2]
    num_rendered, overflow, trap, capacity, P = words[0][:5]
    assert num_rendered == exact[0] and overflow == 0 and trap == 0 and capacity == 10_000_000 and P == 8_000
    assert words[0][6] <= num_rendered and words[0][5] == words[0][6]          # instances after exact culling, all sorted
    with pytest.raises(RuntimeError, match="ticket"):
        _C.header_poll(tickets[0], False)                                       # released by the poll above
    with pytest.raises(RuntimeError, match="ticket"):
        _C.header_poll((1 << 39) + 12345, False)
    assert _C.header_post(out[4]) in tickets                                    # released tickets are handed out again
    # the forward log: an async-mode forward leaves its header in host-visible memory by itself (no copy, no event); the
    # ticket of the last one on this thread costs nothing and may be polled any number of times
    out = _C.rasterize_gaussians(*args, binning_capacity=10_000_000)
    t = _C.last_forward_ticket()
    assert t >= (1 << 40)
    assert _C.header_poll(t, True) == words[0] and _C.header_poll(t, False) == words[0]
    _C.rasterize_gaussians(*args)                                               # an exact-mode forward has no log entry
    assert _C.last_forward_ticket() == -1
    small = _C.rasterize_gaussians(*args, binning_capacity=100)
    assert _C.header_poll(_C.header_post(small[4]), True)[1] == 1               # overflow flag


def test_async_mode_warm_calls_and_overflow_policies(hip_device):
    """warm_calls exact forwards feed the high-water mark.  A view that overflows its async-mode buffer is never handed out or
    differentiated as it is: with the default policy ("verify") the forward waits for the header copy the library posts after
    the scan and renders the view again in exact mode -- image AND gradients are exact mode's, with or without a backward;
    with "drop" / "raise" (nothing waits) the image is incomplete, the gradients are zero -- the backward kernels skip the
    view on the device -- and the deferred check warns / raises.  The capacity is raised either way."""
    import warnings
    from luciddreamer_amd import config
    cloud = synthetic.make_cloud(120_000, "band", 4)          # ~20 k tile instances per view: well above the +4096 slack
    cams = cameras.rotate360_path(384, 256, n_views=8)
    bg = torch.zeros(3)
    g = synthetic.upstream_grad(256, 384)
    config.set_async(False)
    exact = [hp.run_hip(cloud, c, 3, bg, hip_device, g) for c in cams]
    config.reset()
    config.set_async(True, headroom=1.2, check_every=1, warm_calls=len(cams))
    try:
        warm = [hp.run_hip(cloud, c, 3, bg, hip_device, g) for c in cams]          # all exact, mark = max over the path
        key = next(iter(config._hwm))
        assert config._seen[key] == len(cams)
        later = [hp.run_hip(cloud, c, 3, bg, hip_device, g) for c in cams]         # async with enough capacity
        config.drain()
        for a, b, c in zip(exact, warm, later):
            assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["color"], c["color"])
            for k in a["grads"]:
                assert np.array_equal(a["grads"][k], c["grads"][k]), k
        # shrink the mark artificially: the next view overflows.  Default policy: caught inside the forward
        config.set_async(True, headroom=1.0, warm_calls=1)
        config._hwm[key] = 64
        before = config.rerendered_views
        over = hp.run_hip(cloud, cams[0], 3, bg, hip_device, g)
        assert config.rerendered_views == before + 1
        assert np.array_equal(over["color"], exact[0]["color"]) and np.array_equal(over["depth"], exact[0]["depth"])
        for k in exact[0]["grads"]:
            assert np.array_equal(over["grads"][k], exact[0]["grads"][k]), k
        assert config._hwm[key] > 64                                               # raised from the true count
        # ... also when no backward follows and grad mode is on, as in the reference's video loop
        config._hwm[key] = 64
        frame = hp.run_hip(cloud, cams[1], 3, bg, hip_device)
        assert np.array_equal(frame["color"], exact[1]["color"])
        with torch.no_grad():
            config._hwm[key] = 64
            frame = hp.run_hip(cloud, cams[2], 3, bg, hip_device)
        assert np.array_equal(frame["color"], exact[2]["color"])
        again = hp.run_hip(cloud, cams[0], 3, bg, hip_device, g)                   # enough capacity again: async, complete
        assert np.array_equal(again["color"], exact[0]["color"])
        # "drop": nothing waits; the overflowed view's gradients are ZERO (never truncated ones), a warning follows
        config.set_async(True, headroom=1.0, warm_calls=1, on_overflow="drop")
        config._hwm[key] = 64
        dropped = hp.run_hip(cloud, cams[0], 3, bg, hip_device, g)
        assert not np.array_equal(dropped["color"], exact[0]["color"])             # the image it got WAS incomplete
        for k, v in dropped["grads"].items():
            assert float(np.abs(v).max()) == 0.0, k
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            config.drain()
        assert any("binning capacity" in str(x.message) for x in w)
        # "raise"
        config.set_async(True, headroom=1.0, warm_calls=1, on_overflow="raise")
        config._hwm[key] = 64
        hp.run_hip(cloud, cams[0], 3, bg, hip_device, g)
        with pytest.raises(RuntimeError, match="capacity"):
            config.drain()
    finally:
        config.set_async(True)
        config.reset()


def test_fused_grad_accumulation_equals_autograd(hip_device):
    """In-kernel `grad += view gradient` (leaf .grad preallocated) == autograd's dense accumulate, over 3 views."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import config
    cloud = synthetic.make_cloud(15_000, "band", 9)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(256, 144, n_views=6)[:3]]
    g = synthetic.upstream_grad(144, 256).to(hip_device)
    bg = torch.zeros(3, device=hip_device)

    def run(fused):
        leaf = {k: v.to(hip_device).requires_grad_(True) for k, v in cloud.items()}
        m2d = torch.zeros(15_000, 3, device=hip_device, requires_grad=True)
        if fused:
            for t in list(leaf.values()) + [m2d]:
                t.grad = torch.zeros_like(t)
        config.set_fused_grad_accumulation(fused)
        try:
            for c in cams:
                tfx, tfy = hp.tan_fov(c)
                rs = GaussianRasterizationSettings(144, 256, tfx, tfy, bg, 1.0, c.world_view_transform,
                                                   c.full_proj_transform, 3, c.camera_center, False, False)
                col, radii, dep = GaussianRasterizer(rs)(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"],
                                                         shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
                col.backward(g)
        finally:
            config.set_fused_grad_accumulation(False)
        out = {k: v.grad.cpu().numpy() for k, v in leaf.items()}
        out["means2D"] = m2d.grad.cpu().numpy()
        return out

    a, b = run(False), run(True)
    for k in a:
        scale = np.abs(a[k]).max()
        assert scale > 0 and np.abs(a[k] - b[k]).max() <= 2e-5 * scale, k   # float atomics reorder sums run to run


def test_two_stream_view_pipeline_equals_sequential(hip_device):
    """parallel.ViewStreams: forward(i+1) may overlap backward(i) on another stream; the accumulated gradients
    must equal the single-stream result (backward passes are chained by events)."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import config, parallel
    cloud = synthetic.make_cloud(30_000, "band", 4)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(320, 180, n_views=8)]
    g = synthetic.upstream_grad(180, 320).to(hip_device)
    bg = torch.zeros(3, device=hip_device)

    seen_nodes = []

    def run(n_streams, grouped=False, direct=True):
        leaf = {k: v.to(hip_device).requires_grad_(True) for k, v in cloud.items()}
        grads = parallel.FlatGrads(list(leaf.values()))
        m2d = torch.zeros(30_000, 3, device=hip_device, requires_grad=True)
        m2d.grad = torch.zeros_like(m2d)
        rast = []
        for c in cams:
            tfx, tfy = hp.tan_fov(c)
            rast.append(GaussianRasterizer(GaussianRasterizationSettings(
                180, 320, tfx, tfy, bg, 1.0, c.world_view_transform, c.full_proj_transform, 3, c.camera_center,
                False, False)))
        config.set_fused_grad_accumulation(True)
        try:
            pipe = parallel.ViewStreams(hip_device, n_streams, group=3, direct=direct)
            for _ in range(2):                      # two steps: buffers and events are re-used
                grads.zero_()
                m2d.grad.zero_()
                pipe.begin_step()
                for r in rast:
                    def fwd(r=r):
                        out = r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"],
                                shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])[0]
                        seen_nodes.append((grouped, direct, out.grad_fn is not None))
                        return out
                    if grouped:                     # three views per pass of the autograd engine (8 views: 3 + 3 + 2)
                        pipe.run_view(fwd, grad_output=g)
                    else:
                        pipe.run_view(fwd, lambda col: col.backward(g))
                pipe.end_step()
            torch.cuda.synchronize()
        finally:
            config.set_fused_grad_accumulation(False)
        return grads.flat.cpu().numpy().copy(), m2d.grad.cpu().numpy().copy()

    (f1, m1), (f2, m2), (f3, m3), (f4, m4) = run(1), run(2), run(3, grouped=True, direct=False), run(3, grouped=True)
    assert np.abs(f1).max() > 0
    # the per-Gaussian sums run in a fixed order; only the 2-wave LDS adds inside a tile can reorder
    assert np.abs(f1 - f2).max() <= 1e-5 * np.abs(f1).max()
    assert np.abs(m1 - m2).max() <= 1e-5 * np.abs(m1).max()
    # grouped backward passes: accumulation order differs (the engine runs a group's nodes last view first)
    assert np.abs(f1 - f3).max() <= 1e-5 * np.abs(f1).max()
    assert np.abs(m1 - m3).max() <= 1e-5 * np.abs(m1).max()
    # the backward node called directly on the issuing thread (no engine): the per-view order, bit for bit the 2-stream result
    assert np.array_equal(f2, f4) or np.abs(f1 - f4).max() <= 1e-5 * np.abs(f1).max()
    assert np.abs(m1 - m4).max() <= 1e-5 * np.abs(m1).max()
    # ... and that last run took the ONE-CALL path (grad_output known up front, leaves with .grad: forward + backward inside
    # _C.rasterize_view_step, no autograd node: the returned image has no grad_fn); every other run built nodes
    assert all(not has_node for grouped, direct, has_node in seen_nodes if grouped and direct)
    assert all(has_node for grouped, direct, has_node in seen_nodes if not (grouped and direct))
    assert sum(1 for grouped, direct, _ in seen_nodes if grouped and direct) == 16


@pytest.mark.parametrize("one_call", [False, True])
def test_view_pipeline_recovers_views_that_overflow_their_buffer(hip_device, one_call):
    """a ViewStreams step must not lose a view.  The high-water mark is made too small for most views of the
    path (as after a densification, or on a path whose first views are the cheap ones): those views overflow their binning
    buffer, the device-side guard zeroes their gradients, and end_step() runs them again in exact mode -- the step's
    gradients equal the exact-mode sum, `recovered` says how many views it took, nothing is counted as dropped."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import config, parallel
    P = 30_000
    cloud = synthetic.make_cloud(P, "band", 4)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(320, 180, n_views=6)]
    g = synthetic.upstream_grad(180, 320).to(hip_device)
    bg = torch.zeros(3, device=hip_device)
    rast = []
    for c in cams:
        tfx, tfy = hp.tan_fov(c)
        rast.append(GaussianRasterizer(GaussianRasterizationSettings(
            180, 320, tfx, tfy, bg, 1.0, c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)))

    def run(starved):
        leaf = {k: v.to(hip_device).requires_grad_(True) for k, v in cloud.items()}
        grads = parallel.FlatGrads(list(leaf.values()))
        m2d = torch.zeros(P, 3, device=hip_device, requires_grad=True)
        m2d.grad = torch.zeros_like(m2d)
        config.reset()
        config.set_async(True, headroom=1.0, warm_calls=1)
        config.set_fused_grad_accumulation(True)
        config.dropped_views = config.recovered_views = 0
        try:
            fwd = lambda r: r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"],
                              scales=leaf["scales"], rotations=leaf["rotations"])[0]
            if starved:
                with torch.no_grad():
                    fwd(rast[0])                                      # warm call: the key is known ...
                key = next(iter(config._hwm))
                config._hwm[key] = 64                                 # ... and its mark far too small: 64 + 4096 instances
            else:
                config.set_async(False)
            pipe = parallel.ViewStreams(hip_device, 2)
            pipe.begin_step()
            for r in rast:
                if one_call:                     # grad_output up front: forward + backward in one call of the binding; a view
                    pipe.run_view(lambda r=r: fwd(r), grad_output=g)          # that overflowed is re-run through autograd
                else:
                    pipe.run_view(lambda r=r: fwd(r), lambda col: col.backward(g))
            recovered = pipe.end_step()
            torch.cuda.synchronize()
            assert config.current_policy() == "verify"                # the step's temporary policy is gone
        finally:
            config.set_fused_grad_accumulation(False)
            config.reset()
            config.set_async(True)
        return grads.flat.cpu().numpy().copy(), m2d.grad.cpu().numpy().copy(), recovered

    f_exact, m_exact, r0 = run(False)
    f_rec, m_rec, r1 = run(True)
    assert r0 == 0 and r1 >= 1 and config.dropped_views == 0, (r0, r1, config.dropped_views)
    assert np.abs(f_exact).max() > 0
    assert np.abs(f_exact - f_rec).max() <= 1e-5 * np.abs(f_exact).max()
    assert np.abs(m_exact - m_rec).max() <= 1e-5 * np.abs(m_exact).max()


@pytest.mark.parametrize("W,H", [(512, 512), (1920, 1080)])
def test_backward_is_bit_repeatable(hip_device, W, H):
    """a deterministic backward is the default.  512x512 uses the 4-wave (quadrant) shape of the blend
    backward -- every wave sums into its own LDS copy, the copies are added in a fixed order -- and 1080p the 2-wave shape
    (two operands: order-free); the per-Gaussian sums run in a fixed order.  Ten runs, identical bits."""
    cam, cloud = hp.box_setup(60_000, W, H, seed=5, scale_mult=2.0)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.2, 0.1, 0.0])
    first = None
    for _ in range(10):
        out = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
        flat = np.concatenate([out["grads"][k].ravel() for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations")])
        if first is None:
            first = flat
            assert np.abs(flat).max() > 0
        else:
            assert np.array_equal(first.view(np.uint32), flat.view(np.uint32))


def test_view_batch_equals_autograd_accumulation(hip_device):
    """parallel.ViewBatch (lr_views_accumulate: one C call, internal streams) == per-view autograd accumulation."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import parallel
    cloud = synthetic.make_cloud(25_000, "band", 6)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(256, 160, n_views=7)]
    g = synthetic.upstream_grad(160, 256).to(hip_device)
    bg = torch.tensor([0.1, 0.0, 0.2], device=hip_device)
    leaf = {k: v.to(hip_device).requires_grad_(True) for k, v in cloud.items()}
    m2d = torch.zeros(25_000, 3, device=hip_device, requires_grad=True)
    for c in cams:
        tfx, tfy = hp.tan_fov(c)
        rs = GaussianRasterizationSettings(160, 256, tfx, tfy, bg, 1.0, c.world_view_transform, c.full_proj_transform,
                                           3, c.camera_center, False, False)
        col, _, _ = GaussianRasterizer(rs)(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"],
                                           shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
        col.backward(g)
    ref = {k: v.grad.clone() for k, v in leaf.items()}
    ref["means2D"] = m2d.grad.clone()

    acc = {"means3D": torch.zeros_like(leaf["means3D"]), "means2D": torch.zeros_like(m2d),
           "opacity": torch.zeros_like(leaf["opacities"]), "sh": torch.zeros_like(leaf["shs"]),
           "scales": torch.zeros_like(leaf["scales"]), "rotations": torch.zeros_like(leaf["rotations"])}
    batch = parallel.ViewBatch(cams, [g] * len(cams), 3, bg, binning_capacity=400_000, n_streams=2)
    with torch.no_grad():
        for _ in range(2):                           # run twice: workspace / streams / events are re-used
            for t in acc.values():
                t.zero_()
            batch.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(),
                      leaf["rotations"].detach(), leaf["shs"].detach(), acc)
    batch.check()
    names = {"means3D": "means3D", "means2D": "means2D", "opacity": "opacities", "sh": "shs", "scales": "scales",
             "rotations": "rotations"}
    for k, rk in names.items():
        a, b = acc[k].cpu().numpy(), ref[rk].cpu().numpy()
        assert np.abs(b).max() > 0 and np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), k
    # too small a capacity is reported, never written out of bounds
    small = parallel.ViewBatch(cams, [g] * len(cams), 3, bg, binning_capacity=500, n_streams=2)
    small.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(),
              leaf["rotations"].detach(), leaf["shs"].detach(), acc)
    with pytest.raises(RuntimeError, match="capacity"):
        small.check()


@pytest.mark.parametrize("n_views", [1, 2, 5, 7])
def test_view_batch_same_bits_for_every_number_of_chains(hip_device, n_views):
    """lr_views_accumulate spreads a step's views over 1..4 chains -- the chain of the LAST view on the caller's stream, the
    others on streams of the library, the accumulating kernels chained in view order by device-only events, the interleaved
    accumulator zeroed behind the fork, each slot's overflow word reset by its first forward (csrc/api.hip views_core).  With the
    kernel shapes pinned (they follow the number of views in flight otherwise), the sums must not depend on the number of chains
    by a single bit -- twice per batch (streams, events and workspace re-used), from accumulators that start as they are left
    -- and the step must be complete on the caller's stream when the call returns to it (the comparison reads on that stream)."""
    from luciddreamer_amd import _lib, parallel
    P, W, H = 25_000, 256, 160
    cloud = {k: v.to(hip_device) for k, v in synthetic.make_cloud(P, "band", 6).items()}
    cams = [c.to(hip_device) for c in cameras.rotate360_path(W, H, n_views=7)][:n_views]
    g = synthetic.upstream_grad(H, W).to(hip_device)
    bg = torch.tensor([0.1, 0.0, 0.2], device=hip_device)
    shapes = {"means3D": (P, 3), "means2D": (P, 3), "opacity": (P, 1), "sh": (P, 16, 3), "scales": (P, 3), "rotations": (P, 4)}
    _lib.tune_set("blend_quad", 2)
    _lib.tune_set("fwd_pair", 2)
    try:
        want = None
        for n_streams in (1, 2, 3, 4):
            batch = parallel.ViewBatch(cams, [g] * n_views, 3, bg, binning_capacity=400_000, n_streams=n_streams)
            for _ in range(2):
                acc = {k: torch.zeros(s, device=hip_device) for k, s in shapes.items()}
                batch.run(cloud["means3D"], cloud["opacities"], cloud["scales"], cloud["rotations"], cloud["shs"], acc)
                got = {k: v.clone() for k, v in acc.items()}          # on the caller's stream, right behind the call
                batch.check()
                if want is None:
                    want = got
                    assert all(float(v.abs().sum()) > 0 for v in want.values())
                for k in shapes:
                    assert torch.equal(got[k], want[k]), (n_streams, k)
    finally:
        _lib.tune_set("blend_quad", -1)
        _lib.tune_set("fwd_pair", -1)


def test_view_batch_with_fewer_views_than_streams(hip_device):
    """One view on a 3-stream ViewBatch (8 views over 8 GPUs leave 1 view per rank): the overflow words of the unused
    slots of a torch.empty workspace must not be read as garbage by check()."""
    from luciddreamer_amd import parallel
    cloud = {k: v.to(hip_device) for k, v in synthetic.make_cloud(8_000, "band", 3).items()}
    cams = [c.to(hip_device) for c in cameras.rotate360_path(128, 96, n_views=1)]
    g = synthetic.upstream_grad(96, 128).to(hip_device)
    acc = {"means3D": torch.zeros(8_000, 3, device=hip_device), "means2D": torch.zeros(8_000, 3, device=hip_device),
           "opacity": torch.zeros(8_000, 1, device=hip_device), "sh": torch.zeros(8_000, 16, 3, device=hip_device),
           "scales": torch.zeros(8_000, 3, device=hip_device), "rotations": torch.zeros(8_000, 4, device=hip_device)}
    batch = parallel.ViewBatch(cams, [g], 3, torch.zeros(3, device=hip_device), binning_capacity=100_000, n_streams=3)
    P = 8_000
    nbytes = batch.L.lr_views_workspace_bytes(P, 128, 96, 100_000, 3)
    batch._ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=hip_device)      # worst-case stale contents
    batch._ws_key = (P, 100_000)
    batch.run(cloud["means3D"], cloud["opacities"], cloud["scales"], cloud["rotations"], cloud["shs"], acc)
    batch.check()
    assert float(acc["means3D"].abs().max()) > 0


def test_view_batch_with_fused_loss_equals_autograd(hip_device):
    """ViewBatch(targets=...) (lr_views_train_accumulate: render -> L1+DSSIM -> backward per view inside one C call)
    == the autograd op followed by luciddreamer_amd.loss.l1_dssim_loss, summed over the views."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings, GaussianRasterizer
    from luciddreamer_amd import parallel
    from luciddreamer_amd.loss import l1_dssim_loss
    P, W, H = 20_000, 256, 160
    cloud = synthetic.make_cloud(P, "band", 8)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(W, H, n_views=5)]
    gen = torch.Generator().manual_seed(3)
    targets = [torch.rand(3, H, W, generator=gen).to(hip_device) for _ in cams]
    bg = torch.tensor([0.0, 0.1, 0.0], device=hip_device)
    leaf = {k: v.to(hip_device).requires_grad_(True) for k, v in cloud.items()}
    m2d = torch.zeros(P, 3, device=hip_device, requires_grad=True)
    losses = []
    for c, tgt in zip(cams, targets):
        tfx, tfy = hp.tan_fov(c)
        rs = GaussianRasterizationSettings(H, W, tfx, tfy, bg, 1.0, c.world_view_transform, c.full_proj_transform,
                                           3, c.camera_center, False, False)
        col, _, _ = GaussianRasterizer(rs)(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"],
                                           shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
        loss = l1_dssim_loss(col, tgt, 0.2)
        loss.backward()
        losses.append(float(loss.detach()))
    ref = {k: v.grad.clone() for k, v in leaf.items()}
    ref["means2D"] = m2d.grad.clone()

    acc = {"means3D": torch.zeros_like(leaf["means3D"]), "means2D": torch.zeros_like(m2d),
           "opacity": torch.zeros_like(leaf["opacities"]), "sh": torch.zeros_like(leaf["shs"]),
           "scales": torch.zeros_like(leaf["scales"]), "rotations": torch.zeros_like(leaf["rotations"])}
    batch = parallel.ViewBatch(cams, None, 3, bg, binning_capacity=300_000, n_streams=3, targets=targets, lambda_dssim=0.2)
    with torch.no_grad():
        batch.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(),
                  leaf["rotations"].detach(), leaf["shs"].detach(), acc)
    batch.check()
    got = batch.losses.cpu().numpy()
    assert np.abs(got[:, 0] - np.array(losses)).max() <= 1e-6
    names = {"means3D": "means3D", "means2D": "means2D", "opacity": "opacities", "sh": "shs", "scales": "scales",
             "rotations": "rotations"}
    for k, rk in names.items():
        a, b = acc[k].cpu().numpy(), ref[rk].cpu().numpy()
        assert np.abs(b).max() > 0 and np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), k
    with pytest.raises(ValueError):
        parallel.ViewBatch(cams, None, 3, bg, binning_capacity=1000)
