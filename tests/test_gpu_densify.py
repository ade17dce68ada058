"""densify / prune / Adam-state surgery and .ply I/O on the device (luciddreamer_amd.densify,
lr_select_rows / lr_pack_ply_rows) against the plain-torch restatement of the reference's GaussianModel methods
(oracle/densify_oracle.py, CPU).  Pure row movement: results must be bit-exact (the split's new positions /
scales are float arithmetic on the same inputs: 1e-6)."""
import copy
import os
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import densify_oracle as O

pytestmark = pytest.mark.gpu
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "scaling": "_scaling", "rotation": "_rotation"}


class Model:
    """The attributes of the reference's GaussianModel that the densification code touches."""

    def __init__(self, P, dev, seed=0, n_rest=15, with_adam_state=True):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        mk = lambda t: nn.Parameter(t.to(dev).contiguous().requires_grad_(True))
        self._xyz = mk(r(P, 3) * 2)
        self._features_dc = mk(r(P, 1, 3))
        self._features_rest = mk(r(P, n_rest, 3) * 0.1)
        self._opacity = mk(r(P, 1) * 2)
        self._scaling = mk(r(P, 3) * 0.7 - 3.0)
        self._rotation = mk(r(P, 4))
        self.max_sh_degree = 3
        self.percent_dense = 0.01
        groups = [{"params": [getattr(self, a)], "lr": 1e-3, "name": n} for n, a in ATTR.items()]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        if with_adam_state:
            for a in ATTR.values():
                p = getattr(self, a)
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            self.optimizer.step()
            for a in ATTR.values():
                getattr(self, a).grad = None
        self.xyz_gradient_accum = (torch.rand(P, 1, generator=g) * 4e-4).to(dev)
        self.denom = torch.randint(0, 3, (P, 1), generator=g).float().to(dev)
        self.max_radii2D = (torch.rand(P, generator=g) * 40).to(dev)


def to_oracle(model):
    c = lambda t: t.detach().cpu().clone()
    m = {"params": {k: c(getattr(model, a)) for k, a in ATTR.items()}, "percent_dense": model.percent_dense,
         "xyz_gradient_accum": c(model.xyz_gradient_accum), "denom": c(model.denom), "max_radii2D": c(model.max_radii2D)}
    groups = {g["name"]: g for g in model.optimizer.param_groups}
    st0 = model.optimizer.state.get(groups["xyz"]["params"][0], None)
    if st0 is not None and "exp_avg" in st0:
        m["exp_avg"] = {k: c(model.optimizer.state[groups[k]["params"][0]]["exp_avg"]) for k in ATTR}
        m["exp_avg_sq"] = {k: c(model.optimizer.state[groups[k]["params"][0]]["exp_avg_sq"]) for k in ATTR}
    else:
        m["exp_avg"] = None
    return m


def assert_same(model, m, exact=True):
    groups = {g["name"]: g for g in model.optimizer.param_groups}
    for k, a in ATTR.items():
        got, want = getattr(model, a).detach().cpu(), m["params"][k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        if exact or k not in ("xyz", "scaling"):
            assert torch.equal(got, want), k
        else:
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), k
        assert groups[k]["params"][0] is getattr(model, a)            # the model attribute is the optimizer's parameter
        if m.get("exp_avg") is not None:
            st = model.optimizer.state[getattr(model, a)]
            assert torch.equal(st["exp_avg"].cpu(), m["exp_avg"][k]), k
            assert torch.equal(st["exp_avg_sq"].cpu(), m["exp_avg_sq"][k]), k
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(model, s).cpu(), m[s]), s


@pytest.mark.parametrize("P,frac", [(5000, 0.3), (70000, 0.9), (1, 1.0), (2049, 0.0)])
def test_prune_points(hip_device, P, frac):
    from luciddreamer_amd import densify as D
    model = Model(P, hip_device, seed=P)
    m = to_oracle(model)
    mask = torch.rand(P, generator=torch.Generator().manual_seed(1)) < frac
    D.prune_points(model, mask.to(hip_device))
    O.prune_points(m, mask)
    assert_same(model, m)
    # the optimizer keeps working on the re-pointed tensors
    model._xyz.grad = torch.ones_like(model._xyz)
    model.optimizer.step()


def test_clone_split_prune_sequence(hip_device):
    from luciddreamer_amd import densify as D
    P = 20000
    model = Model(P, hip_device, seed=3)
    m = to_oracle(model)
    extent = 5.0
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    g_cpu = grads.cpu().clone()
    D.densify_and_clone(model, grads, 2e-4, extent)
    O.densify_and_clone(m, g_cpu, 2e-4, extent)
    assert_same(model, m)
    assert model._xyz.shape[0] > P
    # split: both sides get the same standard-normal draws
    scaling = torch.exp(m["params"]["scaling"])
    padded = torch.zeros(scaling.shape[0]); padded[:P] = g_cpu.squeeze()
    sel = (padded >= 2e-4) & (scaling.max(dim=1).values > 0.01 * extent)
    stds = scaling[sel].repeat(2, 1)
    samples = torch.randn(stds.shape, generator=torch.Generator().manual_seed(9)) * stds
    D.densify_and_split(model, grads, 2e-4, extent, samples=samples.to(hip_device))
    O.densify_and_split(m, g_cpu, 2e-4, extent, samples=samples)
    assert_same(model, m, exact=False)
    prune = torch.sigmoid(m["params"]["opacity"]).squeeze() < 0.3
    D.prune_points(model, prune.to(hip_device))
    O.prune_points(m, prune)
    assert_same(model, m, exact=False)


def test_degree0_model_with_zero_width_features_rest(hip_device):
    """max_sh_degree = 0: _features_rest is [P, 0, 3] (R/scene/gaussian_model.py:144); prune / clone must work on it."""
    from luciddreamer_amd import densify as D
    P = 5000
    model = Model(P, hip_device, seed=12, n_rest=0)
    model.max_sh_degree = 0
    m = to_oracle(model)
    mask = torch.rand(P, generator=torch.Generator().manual_seed(2)) < 0.4
    D.prune_points(model, mask.to(hip_device))
    O.prune_points(m, mask)
    assert model._features_rest.shape == (int((~mask).sum()), 0, 3)
    assert_same(model, m)
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    D.densify_and_clone(model, grads, 2e-4, 5.0)
    O.densify_and_clone(m, grads.cpu().clone(), 2e-4, 5.0)
    assert_same(model, m)


def test_densify_and_prune_end_to_end(hip_device):
    from luciddreamer_amd import densify as D
    P = 30000
    model = Model(P, hip_device, seed=8)
    m = to_oracle(model)
    torch.manual_seed(123)
    before = model._xyz.shape[0]
    D.densify_and_prune(model, 2e-4, 0.05, 5.0, 20)
    # the split draws its own random numbers here (test_clone_split_prune_sequence feeds both sides the same
    # draws): check the invariants the reference guarantees
    P_new = model._xyz.shape[0]
    assert P_new != before
    for a in ATTR.values():
        assert getattr(model, a).shape[0] == P_new
    for s in ("xyz_gradient_accum", "denom"):
        assert getattr(model, s).shape == (P_new, 1) and not getattr(model, s).any()
    assert model.max_radii2D.shape == (P_new,)
    assert (torch.sigmoid(model._opacity) >= 0.05).all()
    assert (torch.exp(model._scaling).max(dim=1).values <= 0.5).all()
    for a in ATTR.values():
        st = model.optimizer.state[getattr(model, a)]
        assert st["exp_avg"].shape == getattr(model, a).shape
    # training continues: grads flow into the re-pointed parameters
    loss = sum((getattr(model, a) ** 2).sum() for a in ATTR.values())
    loss.backward()
    model.optimizer.step()


def test_no_adam_state_and_patch(hip_device):
    from luciddreamer_amd import densify as D

    class GM(Model):
        pass
    D.patch(GM)
    model = GM(4000, hip_device, seed=2, with_adam_state=False)
    m = to_oracle(model)
    mask = torch.rand(4000, generator=torch.Generator().manual_seed(4)) < 0.5
    model.prune_points(mask.to(hip_device))               # patched method
    O.prune_points(m, mask)
    assert_same(model, m)
    # moments created later by the optimizer are adopted on the next call
    for a in ATTR.values():
        getattr(model, a).grad = torch.ones_like(getattr(model, a))
    model.optimizer.step()
    m = to_oracle(model)
    mask2 = torch.rand(model._xyz.shape[0], generator=torch.Generator().manual_seed(5)) < 0.2
    model.prune_points(mask2.to(hip_device))
    O.prune_points(m, mask2)
    assert_same(model, m)


@pytest.mark.parametrize("n_rest", [15, 0])
def test_save_and_load_ply(hip_device, tmp_path, n_rest):
    from luciddreamer_amd import densify as D
    P = 3000
    model = Model(P, hip_device, seed=6, n_rest=n_rest)
    model.max_sh_degree = 3 if n_rest == 15 else 0
    path = os.path.join(tmp_path, "pc", "point_cloud.ply")
    D.save_ply(model, path)
    rows = O.ply_rows(to_oracle(model)).numpy()
    v = D.read_ply(path)
    names = D.ply_attribute_names(n_rest)
    assert list(v.keys()) == names                                   # construct_list_of_attributes order
    got = np.stack([v[k] for k in names], axis=1)
    assert np.array_equal(got, rows)
    other = types.SimpleNamespace(max_sh_degree=model.max_sh_degree)
    D.load_ply(other, path, device=hip_device)
    for a in ATTR.values():
        assert torch.equal(getattr(other, a).detach(), getattr(model, a).detach()), a


def test_select_rows_rejects_bad_arguments(hip_device):
    import ctypes
    from luciddreamer_amd import _lib
    L = _lib.lib()
    a = torch.zeros(10, 3, device=hip_device)
    cnt = torch.zeros(1, dtype=torch.int32, device=hip_device)
    ws = torch.empty(L.lr_select_workspace_bytes(10), dtype=torch.uint8, device=hip_device)
    mask = torch.ones(10, dtype=torch.uint8, device=hip_device)
    arr = lambda *p: (ctypes.c_void_p * len(p))(*p)
    rb = (ctypes.c_uint * 1)(12)
    s = torch.cuda.current_stream(hip_device).cuda_stream
    # in-place compaction is refused
    assert L.lr_select_rows(10, mask.data_ptr(), 1, arr(a.data_ptr()), arr(a.data_ptr()), rb, 0, cnt.data_ptr(),
                            ws.data_ptr(), ws.numel(), s) == _lib.LR_ERR_INVALID_ARG
    rb_bad = (ctypes.c_uint * 1)(6)
    b = torch.zeros(10, 3, device=hip_device)
    assert L.lr_select_rows(10, mask.data_ptr(), 1, arr(a.data_ptr()), arr(b.data_ptr()), rb_bad, 0, cnt.data_ptr(),
                            ws.data_ptr(), ws.numel(), s) == _lib.LR_ERR_INVALID_ARG


def test_add_densification_stats(hip_device):
    from luciddreamer_amd import densify as D
    P = 30000
    m = Model(P, hip_device, seed=5)
    g = torch.Generator().manual_seed(6)
    radii = torch.randint(-1, 40, (P,), generator=g, dtype=torch.int32).clamp_min(0).to(hip_device)
    vs = torch.zeros(P, 3, device=hip_device, requires_grad=True)
    vs.grad = torch.randn(P, 3, generator=g).to(hip_device)
    acc0, den0, mr0 = m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()
    D.add_densification_stats(m, vs, radii)
    vis = radii > 0
    mr0[vis] = torch.max(mr0[vis], radii[vis].float())                                   # luciddreamer.py:310-311
    acc0[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)                     # gaussian_model.py:406
    den0[vis] += 1                                                                     # :407
    assert torch.equal(m.max_radii2D, mr0) and torch.equal(m.denom, den0)
    assert torch.allclose(m.xyz_gradient_accum, acc0, rtol=1e-6, atol=0)
