"""FusedAdam (lr_adam_step) against torch.optim.Adam on the parameter set of a GaussianModel: same updates over many
steps (float32, a few ulps), same state layout, works with the device-side densification."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}


def _params(P, dev, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: nn.Parameter(torch.randn((P,) + s, generator=g).to(dev)) for k, s in SHAPES.items()}


def test_matches_torch_adam(hip_device):
    from luciddreamer_amd.optim import FusedAdam
    P = 50_000
    a, b = _params(P, hip_device, 0), _params(P, hip_device, 0)
    groups = lambda d: [{"params": [d[k]], "lr": LRS[k], "name": k} for k in SHAPES]
    ref = torch.optim.Adam(groups(a), lr=0.0, eps=1e-15)
    fus = FusedAdam(groups(b), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(1)
    for it in range(25):
        for k in SHAPES:
            grad = (torch.randn((P,) + SHAPES[k], generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(hip_device)
            if it % 5 == 0:
                grad[: P // 2] = 0                                  # rows that were not visible
            a[k].grad = grad.clone()
            b[k].grad = grad.clone()
        if it == 10:
            for grp in list(ref.param_groups) + list(fus.param_groups):     # update_learning_rate
                if grp["name"] == "xyz":
                    grp["lr"] = 1.0e-4
        ref.step()
        fus.step()
    for k in SHAPES:
        close = lambda x, y: (x - y).abs().max().item() <= 2e-6 * y.abs().max().item()
        assert close(b[k], a[k]), k
        sa, sb = ref.state[a[k]], fus.state[b[k]]
        assert int(sa["step"]) == int(sb["step"]) == 25
        assert close(sb["exp_avg"], sa["exp_avg"]) and close(sb["exp_avg_sq"], sa["exp_avg_sq"]), k


def test_odd_sizes_and_unaligned_tensors_take_the_same_steps(hip_device):
    """The kernel moves 16 bytes per lane where all four arrays are 16-byte aligned and finishes the tail (and whole tensors that
    are not aligned: a parameter that is a view at an odd offset) one float at a time; every element must take the step torch
    takes, whichever path it is on."""
    from luciddreamer_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    sizes = [1, 3, 5, 1021, 4096 + 7]
    base = [torch.randn(n + 1, generator=g).to(hip_device) for n in sizes]
    a = [nn.Parameter(t[:-1].clone()) for t in base]                       # aligned, numel % 4 != 0
    b = [nn.Parameter(t[:-1].clone()) for t in base]
    ua, ub = nn.Parameter(base[-1].clone()[1:]), nn.Parameter(base[-1].clone()[1:])     # 4 bytes off a 16-byte boundary
    assert ub.data_ptr() % 16 == 4 and ub.is_contiguous()
    ref = torch.optim.Adam(a + [ua], lr=1e-2, eps=1e-15)
    fus = FusedAdam(b + [ub], lr=1e-2, eps=1e-15)
    for it in range(7):
        for x, y in zip(a + [ua], b + [ub]):
            gr = torch.randn(x.shape, generator=g).to(hip_device)
            x.grad, y.grad = gr.clone(), gr.clone()
        ref.step()
        fus.step()
    for x, y in zip(a + [ua], b + [ub]):
        assert (x - y).abs().max().item() <= 2e-6 * max(x.abs().max().item(), 1e-12), x.shape
        assert (ref.state[x]["exp_avg_sq"] - fus.state[y]["exp_avg_sq"]).abs().max().item() <= 2e-6 * ref.state[x]["exp_avg_sq"].abs().max().item()


def test_params_without_grad_are_skipped_and_cpu_is_rejected(hip_device):
    from luciddreamer_amd.optim import FusedAdam
    p = _params(100, hip_device, 3)
    opt = FusedAdam([{"params": [p[k]], "lr": LRS[k], "name": k} for k in SHAPES], lr=0.0, eps=1e-15)
    before = p["rotation"].detach().clone()
    p["xyz"].grad = torch.ones_like(p["xyz"])
    opt.step()
    assert torch.equal(p["rotation"], before) and len(opt.state[p["rotation"]]) == 0
    q = nn.Parameter(torch.zeros(4, 3))
    q.grad = torch.ones(4, 3)
    with pytest.raises(RuntimeError):
        FusedAdam([q], lr=1e-3).step()


def test_works_with_device_densification(hip_device):
    from luciddreamer_amd import densify as D
    from luciddreamer_amd.optim import FusedAdam
    from tests.test_gpu_densify import Model, ATTR
    m = Model(5000, hip_device, seed=1, with_adam_state=False)
    m.optimizer = FusedAdam([{"params": [getattr(m, a)], "lr": 1e-3, "name": n} for n, a in ATTR.items()], lr=0.0, eps=1e-15)
    for a in ATTR.values():
        getattr(m, a).grad = torch.ones_like(getattr(m, a))
    m.optimizer.step()
    mask = torch.rand(5000, generator=torch.Generator().manual_seed(2)) < 0.4
    keep = ~mask
    want = m.optimizer.state[m._xyz]["exp_avg"][keep.to(hip_device)].clone()
    D.prune_points(m, mask.to(hip_device))
    assert torch.equal(m.optimizer.state[m._xyz]["exp_avg"], want)
    for a in ATTR.values():
        getattr(m, a).grad = torch.ones_like(getattr(m, a))
    m.optimizer.step()                                        # moments follow the re-pointed parameters
    assert int(m.optimizer.state[m._xyz]["step"]) == 2


def test_coefficients_above_the_active_sh_degree_stay_untouched_and_equal_torch(hip_device):
    """LucidDreamer raises the SH degree every 1000 of its 2990 iterations: the coefficients above the active degree have zero
    gradient and zero moments throughout, and lr_adam_step skips their stores (adam.hip: the update of such an element is the
    identity, signed zeros included).  Against torch.optim.Adam over 12 steps with the degree going 0 -> 1 on the way: bit-equal
    parameters where nothing ever arrived, the usual few ulps elsewhere, and moments that start from zero when a band wakes up."""
    from luciddreamer_amd.optim import FusedAdam
    P = 20_000
    a, b = _params(P, hip_device, 3), _params(P, hip_device, 3)
    with torch.no_grad():
        for d in (a, b):
            d["f_rest"][:, 7:, 1] = -0.0                                       # signed zeros among the parameters themselves
    start = b["f_rest"].detach().clone()
    groups = lambda d: [{"params": [d[k]], "lr": LRS[k], "name": k} for k in SHAPES]
    ref = torch.optim.Adam(groups(a), lr=0.0, eps=1e-15)
    fus = FusedAdam(groups(b), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(4)
    for it in range(12):
        active = 0 if it < 6 else 3                                            # coefficients of f_rest that receive a gradient
        for k in SHAPES:
            grad = torch.randn((P,) + SHAPES[k], generator=g).to(hip_device)
            if k == "f_rest":
                grad[:, active:, :] = 0
            a[k].grad, b[k].grad = grad.clone(), grad.clone()
        ref.step()
        fus.step()
    assert torch.equal(b["f_rest"][:, 3:, :], start[:, 3:, :])                 # never touched: the same bits, -0.0 included
    assert torch.equal(a["f_rest"][:, 3:, :], start[:, 3:, :])                 # ... as with torch's own arithmetic
    for k in SHAPES:
        close = lambda x, y: (x - y).abs().max().item() <= 2e-6 * max(y.abs().max().item(), 1e-30)
        assert close(b[k], a[k]), k
        sa, sb = ref.state[a[k]], fus.state[b[k]]
        assert close(sb["exp_avg"], sa["exp_avg"]) and close(sb["exp_avg_sq"], sa["exp_avg_sq"]), k
    assert float(fus.state[b["f_rest"]]["exp_avg_sq"][:, 3:, :].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# The armed pair: backward without zero-fill + masked step (LR_ACC_NO_ZERO_FILL + lr_adam_step_masked; optim.FusedAdam.arm_fused_backward)
# ---------------------------------------------------------------------------------------------------------------------
def _two_clouds(P, dev, seed=7):
    from luciddreamer_amd import synthetic
    from luciddreamer_amd.gaussian_renderer import GaussianCloud
    c = synthetic.make_cloud(P, "band", seed)
    mk = lambda: GaussianCloud(c["means3D"].to(dev), c["scales"].to(dev), c["rotations"].to(dev), c["opacities"].to(dev),
                               c["shs"].to(dev), active_sh_degree=0)
    return mk(), mk()


def _adam_for(cloud):
    from luciddreamer_amd.optim import FusedAdam
    named = {"xyz": cloud._xyz, "f_dc": cloud._features_dc, "f_rest": cloud._features_rest, "opacity": cloud._opacity,
             "scaling": cloud._scaling, "rotation": cloud._rotation}
    return FusedAdam([{"params": [nn.Parameter(v) if not isinstance(v, nn.Parameter) else v], "lr": LRS[k], "name": k}
                      for k, v in named.items()], lr=0.0, eps=1e-15)


@pytest.mark.parametrize("W,H", [(320, 192), (1280, 720)])
def test_step_taken_by_the_backward_gives_the_bits_of_backward_plus_step(hip_device, W, H):
    """Same cloud twice, same views, same upstream gradients.  A: raw-mode backward writes the gradients, FusedAdam.step()
    applies them (lr_backward_raw + lr_adam_step).  B: the optimizer is armed, the backward writes only the rows of the Gaussians
    the view touches into tensors autograd never sees (LR_ACC_NO_ZERO_FILL) and step() takes every other gradient as zero without
    reading it (lr_adam_step_masked).  After every iteration all six parameter tensors and both moments must be the SAME BITS, and so must
    the screen-space gradients the densification statistics read -- over views that see a fraction of the band cloud (most
    rows are 'the rest'), with the SH degree raised and a learning rate changed on the way, and an un-armed iteration in between."""
    from luciddreamer_amd import cameras, synthetic
    from luciddreamer_amd.gaussian_renderer import render_raw
    P = 40_000
    a, b = _two_clouds(P, hip_device)
    for cl in (a, b):
        for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            setattr(cl, n, nn.Parameter(getattr(cl, n).detach()))
    opt_a, opt_b = _adam_for(a), _adam_for(b)
    cams = [c.to(hip_device) for c in cameras.rotate360_path(W, H, n_views=12)]
    bg = torch.zeros(3, device=hip_device)
    gen = torch.Generator().manual_seed(5)
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    visited = []
    for it in range(10):
        cam = cams[(5 * it) % 12]
        if it == 4:
            a.active_sh_degree = b.active_sh_degree = 2
        if it == 6:
            for opt in (opt_a, opt_b):
                for grp in opt.param_groups:
                    if grp["name"] == "xyz":
                        grp["lr"] = 1.0e-4
        g = torch.randn(3, H, W, generator=gen).to(hip_device)
        armed = it != 7                                              # one iteration the plain way in the middle of the run
        pa = render_raw(cam, a, bg_color=bg)
        (pa["render"] * g).sum().backward()
        opt_a.step()
        opt_a.zero_grad(set_to_none=True)
        if armed:
            assert opt_b.arm_fused_backward()
            # the armed backward's gradient tensors are NOT zero-filled: poison the allocator's cache, so that whatever the
            # backward leaves unwritten and the masked step then reads is NaN, not the zeros of fresh memory
            junk = [torch.full_like(getattr(b, n), float("nan")) for n in names]
            del junk
        pb = render_raw(cam, b, bg_color=bg)
        (pb["render"] * g).sum().backward()
        if armed:
            assert all(getattr(b, n).grad is None for n in names) and opt_b._fused_pending is not None
        else:
            assert all(getattr(b, n).grad is not None for n in names)
        opt_b.step()
        opt_b.zero_grad(set_to_none=True)
        assert opt_b._fused_pending is None
        visited.append(float((pb["radii"] > 0).float().mean()))
        assert torch.equal(pa["viewspace_points"].grad, pb["viewspace_points"].grad), it
        for n in names:
            x, y = getattr(a, n), getattr(b, n)
            assert torch.equal(x, y), (it, n, float((x - y).abs().max()))
            sa, sb = opt_a.state[x], opt_b.state[y]
            assert int(sa["step"]) == int(sb["step"]) == it + 1
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), (it, n)
    assert 0.02 < min(visited) and max(visited) < 0.6, visited      # the views really split the cloud into visited rows and the rest
    assert float((a._xyz.detach() - _two_clouds(P, hip_device)[0]._xyz.detach()).abs().max()) > 0


def test_fused_step_refuses_what_it_cannot_do_exactly(hip_device):
    """Armed, but ... (i) the parameter set is replaced between backward and step(), (ii) a parameter receives a second gradient:
    step() raises instead of applying half a step silently; (iii) the backward runs over OTHER tensors than the optimizer's: the
    offer is simply not taken (plain gradients, plain step); (iv) an armed backward that never ran is forgotten by step()."""
    from luciddreamer_amd import cameras
    from luciddreamer_amd.gaussian_renderer import render_raw
    P, W, H = 5000, 160, 96
    cam = cameras.rotate360_path(W, H, n_views=4)[1].to(hip_device)
    g = torch.ones(3, H, W, device=hip_device)

    def fresh():
        c, other = _two_clouds(P, hip_device, seed=9)
        for cl in (c, other):
            for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
                setattr(cl, n, nn.Parameter(getattr(cl, n).detach()))
        return c, other, _adam_for(c)
    c, other, opt = fresh()
    assert opt.arm_fused_backward()
    (render_raw(cam, c)["render"] * g).sum().backward()
    opt.param_groups[0]["params"][0] = nn.Parameter(c._xyz.detach().clone())       # (i) what a densification does
    with pytest.raises(RuntimeError, match="parameter set changed"):
        opt.step()
    c, other, opt = fresh()
    assert opt.arm_fused_backward()
    (render_raw(cam, c)["render"] * g).sum().backward()
    c._opacity.grad = torch.ones_like(c._opacity)                                    # (ii)
    with pytest.raises(RuntimeError, match="received a .grad"):
        opt.step()
    c, other, opt = fresh()
    assert opt.arm_fused_backward()
    (render_raw(cam, other)["render"] * g).sum().backward()                           # (iii) not this optimizer's tensors
    assert other._xyz.grad is not None and opt._fused_pending is None
    before = c._xyz.detach().clone()
    opt.step()                                                                        # nothing to do: c has no gradients
    assert torch.equal(c._xyz, before)
    assert opt.arm_fused_backward()                                                   # (iv)
    opt.step()
    (render_raw(cam, c)["render"] * g).sum().backward()                               # the stale offer is gone: plain gradients
    assert c._xyz.grad is not None
