"""CPU: the mechanics of luciddreamer_amd.install() / uninstall() -- what is replaced where, re-binding of names a caller
imported before the call, fall-through to the originals for calls the fused pieces do not cover, restoration.  The fused
pieces themselves run on the GPU (tests/test_gpu_reference_stack.py::test_install_switches_the_unchanged_loop_onto_the_fused_pieces)."""
import sys
import types

import pytest
import torch

import luciddreamer_amd
from luciddreamer_amd import dropin


def _fake_reference():
    gr = types.ModuleType("fake_gaussian_renderer")
    calls = []

    def render(viewpoint_camera, pc, opt, bg_color, scaling_modifier=1.0, override_color=None, render_only=False):
        calls.append("render")
        return {"render": torch.zeros(3, 4, 4)}
    gr.render = render
    ls = types.ModuleType("fake_loss")
    ls.l1_loss = lambda a, b: torch.abs(a - b).mean()
    ls.ssim = lambda a, b, window_size=11, size_average=True: torch.tensor(1.0)

    class GaussianModel:
        def training_setup(self, args):
            self.optimizer = torch.optim.Adam([{"params": [torch.nn.Parameter(torch.zeros(3))], "lr": 0.1, "name": "xyz"}],
                                              lr=0.0, eps=1e-15)

        def add_densification_stats(self, vsp, flt):
            calls.append("stats")

        def densify_and_prune(self, *a):
            calls.append("densify")
    gm = types.ModuleType("fake_gaussian_model")
    gm.GaussianModel = GaussianModel
    caller = types.ModuleType("fake_luciddreamer")            # `from gaussian_renderer import render` etc., done BEFORE install()
    caller.render, caller.l1_loss, caller.ssim = gr.render, ls.l1_loss, ls.ssim
    for m in (gr, ls, gm, caller):
        sys.modules[m.__name__] = m
    return gr, ls, gm, caller, calls


def test_install_replaces_rebinds_falls_through_and_uninstall_restores():
    gr, ls, gm, caller, calls = _fake_reference()
    try:
        orig = (gr.render, ls.l1_loss, ls.ssim, gm.GaussianModel.training_setup, gm.GaussianModel.add_densification_stats,
                gm.GaussianModel.densify_and_prune)
        assert torch.autograd.is_multithreading_enabled()
        # "auto" (the default) leaves the PROCESS-WIDE autograd setting alone when the caller has threads of its own
        import threading
        stop = threading.Event()
        t = threading.Thread(target=stop.wait, daemon=True)
        t.start()
        h0 = luciddreamer_amd.install(gr, ls, gm)
        assert torch.autograd.is_multithreading_enabled() and not h0.backward_on_calling_thread
        luciddreamer_amd.uninstall(h0)
        stop.set()
        t.join()
        h = luciddreamer_amd.install(gr, ls, gm, backward_on_calling_thread=True)
        assert not torch.autograd.is_multithreading_enabled()                         # backward on the calling thread while installed
        assert gr.render is not orig[0] and caller.render is gr.render               # re-bound in the module that imported it by name
        assert caller.l1_loss is ls.l1_loss and caller.ssim is ls.ssim and ls.l1_loss is not orig[1]
        assert gm.GaussianModel.densify_and_prune is not orig[5] and hasattr(gm.GaussianModel, "save_ply")
        # a call the fused render does not cover (CPU tensors, no stored parameters) goes to the original
        pc = types.SimpleNamespace()
        opt = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
        assert "render" in gr.render(None, pc, opt, torch.zeros(3)) and calls == ["render"]
        # l1 of something that is not an RGB image on the device: the reference's own expression
        a, b = torch.rand(1, 8, 8), torch.rand(1, 8, 8)
        assert torch.equal(ls.l1_loss(a, b), torch.abs(a - b).mean())
        # ssim the shared pass does not cover (another window, host tensors): the function install() replaced answers
        assert float(ls.ssim(a, b, window_size=7)) == 1.0 and float(ls.ssim(torch.rand(3, 8, 8), torch.rand(3, 8, 8))) == 1.0
        # Adam: parameters on the host -> the reference's optimizer is kept (FusedAdam has no CPU path)
        m = gm.GaussianModel()
        m.training_setup(None)
        assert type(m.optimizer) is torch.optim.Adam
        # statistics without the radii the fused render attaches -> the original method
        m.add_densification_stats(types.SimpleNamespace(grad=None), torch.zeros(3, dtype=torch.bool))
        assert calls[-1] == "stats"
        luciddreamer_amd.uninstall(h)
        assert torch.autograd.is_multithreading_enabled()
        now = (gr.render, ls.l1_loss, ls.ssim, gm.GaussianModel.training_setup, gm.GaussianModel.add_densification_stats,
               gm.GaussianModel.densify_and_prune)
        assert all(x is y for x, y in zip(orig, now)) and caller.render is orig[0] and caller.ssim is orig[2]
        assert not hasattr(gm.GaussianModel, "save_ply")
    finally:
        for m in (gr, ls, gm, caller):
            sys.modules.pop(m.__name__, None)


def test_install_takes_a_namespace_and_switches_individually():
    gr, ls, gm, caller, _ = _fake_reference()
    try:
        ns = types.SimpleNamespace(gaussian_renderer=gr, loss=ls, gaussian_model=gm)
        orig_render, orig_l1 = gr.render, ls.l1_loss
        h = dropin.install(ns, losses=False, densify=False)
        assert gr.render is not orig_render and ls.l1_loss is orig_l1 and not hasattr(gm.GaussianModel, "save_ply")
        dropin.uninstall(h)
        assert gr.render is orig_render
    finally:
        for m in (gr, ls, gm, caller):
            sys.modules.pop(m.__name__, None)


def test_visibility_filter_keeps_the_masked_max_radii_update_on_the_device_and_is_a_plain_mask_otherwise():
    """The filter the installed render() returns (dropin._VisFilter): the loop's own
    `max_radii2D[f] = torch.max(max_radii2D[f], radii[f])` (R/luciddreamer.py:310-312) runs as one dense where() -- no nonzero, no
    host read -- with the values of the eager statement; every other use of the filter or of a selection taken with it is the
    plain operation (the reference's own add_densification_stats body, arithmetic, scalars, pickling)."""
    import pickle
    from luciddreamer_amd import dropin
    torch.manual_seed(0)
    P = 257
    radii = torch.randint(0, 20, (P,), dtype=torch.int32)
    mask = radii > 5
    f = dropin._VisFilter.wrap(mask.clone(), radii)
    mr = torch.rand(P) * 10
    want, got = mr.clone(), mr.clone()
    want[mask] = torch.max(want[mask], radii[mask])
    n0 = dropin.lazy_assignments
    got[f] = torch.max(got[f], radii[f])
    assert torch.equal(want, got) and dropin.lazy_assignments == n0 + 1 and got.dtype == torch.float32
    # [P, 1] destination, torch.maximum spelling
    d_want, d_got, ones = torch.zeros(P, 1), torch.zeros(P, 1), torch.rand(P, 1)
    d_want[mask] = torch.maximum(d_want[mask], ones[mask])
    d_got[f] = torch.maximum(d_got[f], ones[f])
    assert torch.equal(d_want, d_got) and dropin.lazy_assignments == n0 + 2
    # the reference's own statistics body (scene/gaussian_model.py:405-407): getitem, in-place add, setitem; tuple index
    acc_want, acc_got, g = torch.zeros(P, 1), torch.zeros(P, 1), torch.rand(P, 3)
    acc_want[mask] += torch.norm(g[mask, :2], dim=-1, keepdim=True)
    acc_got[f] += torch.norm(g[f, :2], dim=-1, keepdim=True)
    assert torch.equal(acc_want, acc_got)
    # a selection used in any other way is the eager selection
    sel = mr[f]
    assert sel.shape == mr[mask].shape and torch.equal(sel * 2, mr[mask] * 2) and float(sel.sum()) == float(mr[mask].sum())
    assert type(sel * 2) is torch.Tensor and torch.equal(torch.max(mr[f], mr[mask]), mr[mask])
    x, y = mr.clone(), mr.clone()
    x[f] = 5.0
    y[mask] = 5.0
    assert torch.equal(x, y)
    x[f] = mr[f] * 2
    y[mask] = mr[mask] * 2
    assert torch.equal(x, y)
    # the filter itself: a bool tensor in every other respect, and it pickles as one
    assert int(f.sum()) == int(mask.sum()) and type(~f) is torch.Tensor and torch.equal(~f, ~mask)
    back = pickle.loads(pickle.dumps(f))
    assert type(back) is torch.Tensor and torch.equal(back, mask)
    # a selection that outlives an in-place change of its source is refused, not silently different from eager indexing
    src = mr.clone()
    held = src[f]
    src.add_(1.0)
    with pytest.raises(RuntimeError, match="changed in place"):
        held.sum()


def test_plain_iteration_follows_the_reference_loop_schedule():
    """install(fuse_step=True) arms the optimizer on exactly the iterations on which the reference's loop
    (R/luciddreamer.py:305-327) runs loss.backward() and optimizer.step() back to back on an unchanged parameter set.  The
    predicate against a literal walk through that loop's control flow, for the reference's own GSParams and three variations."""
    import types
    from luciddreamer_amd.dropin import plain_iteration

    def walk(a, iteration):
        """(did the loop touch the parameter set before the step, did it step) as luciddreamer.py:305-327 decides them"""
        touched = False
        if iteration < a.densify_until_iter:
            if iteration > a.densify_from_iter and iteration % a.densification_interval == 0:
                touched = True                                   # densify_and_prune
            if iteration % a.opacity_reset_interval == 0 or (a.white_background and iteration == a.densify_from_iter):
                touched = True                                   # reset_opacity
        stepped = iteration < a.iterations
        return touched, stepped

    base = dict(iterations=2990, densify_until_iter=15_000, densify_from_iter=500, densification_interval=100,
                opacity_reset_interval=3000, white_background=False)
    for change in ({}, {"white_background": True}, {"densify_until_iter": 1200, "opacity_reset_interval": 700},
                   {"iterations": 301, "densify_from_iter": 5, "densification_interval": 10}):
        a = types.SimpleNamespace(**dict(base, **change))
        armed = 0
        for it in range(1, a.iterations + 1):
            touched, stepped = walk(a, it)
            assert plain_iteration(a, it) == (stepped and not touched), (change, it)
            armed += plain_iteration(a, it)
        assert 0 < armed < a.iterations
    # a training_args object without the schedule's fields never arms
    assert not plain_iteration(types.SimpleNamespace(iterations=10), 3)
