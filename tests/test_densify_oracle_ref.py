"""Pins oracle/densify_oracle.py (the checker of luciddreamer_amd.densify) to the reference's own
GaussianModel (R/scene/gaussian_model.py:176-403), imported unchanged and run on CPU tensors (oracle/ref_python.py maps
its hard-coded device="cuda" to the CPU).  Row surgery is pure data movement and the split's arithmetic is the same
torch expression on the same inputs, so every tensor must be bit-identical -- parameters, both Adam moments, the three
statistics tensors -- after prune, clone, split and the full densify_and_prune, with and without optimizer state."""
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import densify_oracle as O
from oracle import ref_python as rp

pytestmark = pytest.mark.skipif(not rp.available(), reason="reference Python sources not present")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "scaling": "_scaling", "rotation": "_rotation"}


@pytest.fixture()
def ref():
    captured = {}
    ply = types.ModuleType("plyfile")               # plyfile is not installed: capture what save_ply hands to it

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements, name
            return elements

    class PlyData:
        def __init__(self, els):
            self.els = els

        def write(self, path):
            captured["path"] = path
    ply.PlyElement, ply.PlyData = PlyElement, PlyData
    had = sys.modules.get("plyfile")
    sys.modules["plyfile"] = ply
    try:
        with rp.reference_modules("port") as R, rp.cuda_as_cpu():
            R.captured = captured
            yield R
    finally:
        if had is None:
            sys.modules.pop("plyfile", None)
        else:
            sys.modules["plyfile"] = had


def make_reference_model(R, P, seed, with_adam_state=True, n_rest=15):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    gm = R.gaussian_model.GaussianModel(3)
    mk = lambda t: nn.Parameter(t.contiguous().requires_grad_(True))
    gm._xyz, gm._features_dc, gm._features_rest = mk(r(P, 3) * 2), mk(r(P, 1, 3)), mk(r(P, n_rest, 3) * 0.1)
    gm._opacity, gm._scaling, gm._rotation = mk(r(P, 1) * 2), mk(r(P, 3) * 0.7 - 3.0), mk(r(P, 4))
    gm.spatial_lr_scale = 1.0
    gm.training_setup(R.arguments.GSParams())                        # the reference's own Adam groups (:151-169)
    if with_adam_state:
        for a in ATTR.values():
            getattr(gm, a).grad = torch.randn(getattr(gm, a).shape, generator=g)
        for grp in gm.optimizer.param_groups:
            grp["lr"] = 1e-3
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
    gm.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 4e-4
    gm.denom = torch.randint(0, 3, (P, 1), generator=g).float()
    gm.max_radii2D = torch.rand(P, generator=g) * 40
    return gm


def to_oracle(gm):
    c = lambda t: t.detach().clone()
    m = {"params": {k: c(getattr(gm, a)) for k, a in ATTR.items()}, "percent_dense": gm.percent_dense,
         "xyz_gradient_accum": c(gm.xyz_gradient_accum), "denom": c(gm.denom), "max_radii2D": c(gm.max_radii2D)}
    st = gm.optimizer.state
    if len(st):
        m["exp_avg"] = {k: c(st[getattr(gm, a)]["exp_avg"]) for k, a in ATTR.items()}
        m["exp_avg_sq"] = {k: c(st[getattr(gm, a)]["exp_avg_sq"]) for k, a in ATTR.items()}
    else:
        m["exp_avg"] = None
    return m


def assert_same(gm, m):
    for k, a in ATTR.items():
        got, want = getattr(gm, a).detach(), m["params"][k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert torch.equal(got, want), k
        if m.get("exp_avg") is not None:
            assert torch.equal(gm.optimizer.state[getattr(gm, a)]["exp_avg"], m["exp_avg"][k]), k
            assert torch.equal(gm.optimizer.state[getattr(gm, a)]["exp_avg_sq"], m["exp_avg_sq"][k]), k
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(gm, s), m[s]), s


@pytest.mark.parametrize("with_state", [True, False])
def test_prune_points(ref, with_state):
    gm = make_reference_model(ref, 3000, 1, with_state)
    m = to_oracle(gm)
    mask = torch.rand(3000, generator=torch.Generator().manual_seed(5)) < 0.3
    gm.prune_points(mask)
    O.prune_points(m, mask)
    assert gm.get_xyz.shape[0] == int((~mask).sum())
    assert_same(gm, m)


@pytest.mark.parametrize("with_state", [True, False])
def test_clone_then_split(ref, with_state):
    gm = make_reference_model(ref, 4000, 2, with_state)
    m = to_oracle(gm)
    grads = gm.xyz_gradient_accum / gm.denom
    grads[grads.isnan()] = 0.0
    gm.densify_and_clone(grads, 2e-4, 3.0)
    O.densify_and_clone(m, grads.clone(), 2e-4, 3.0)
    assert gm.get_xyz.shape[0] > 4000
    assert_same(gm, m)
    torch.manual_seed(11)
    gm.densify_and_split(grads, 2e-4, 3.0)
    torch.manual_seed(11)
    O.densify_and_split(m, grads.clone(), 2e-4, 3.0)
    assert_same(gm, m)


@pytest.mark.parametrize("screen", [None, 20])
@pytest.mark.parametrize("seed", [3, 4])
def test_densify_and_prune(ref, seed, screen):
    gm = make_reference_model(ref, 5000, seed)
    m = to_oracle(gm)
    P0 = gm.get_xyz.shape[0]
    torch.manual_seed(100 + seed)
    gm.densify_and_prune(2e-4, 0.005, 3.0, screen)
    torch.manual_seed(100 + seed)
    O.densify_and_prune(m, 2e-4, 0.005, 3.0, screen)
    assert gm.get_xyz.shape[0] != P0
    assert_same(gm, m)
    # a second round on the result (statistics were reset by the first: no clones/splits, opacity prune only)
    gm.xyz_gradient_accum += 1e-3
    gm.denom += 1
    m["xyz_gradient_accum"] += 1e-3
    m["denom"] += 1
    torch.manual_seed(7)
    gm.densify_and_prune(2e-4, 0.005, 3.0, screen)
    torch.manual_seed(7)
    O.densify_and_prune(m, 2e-4, 0.005, 3.0, screen)
    assert_same(gm, m)


def test_save_ply_rows_and_attribute_names(ref, tmp_path):
    gm = make_reference_model(ref, 257, 9)
    m = to_oracle(gm)
    gm.save_ply(str(tmp_path / "x.ply"))
    el = ref.captured["elements"]
    names = el.dtype.names
    assert names[:6] == ("x", "y", "z", "nx", "ny", "nz") and names[-4:] == ("rot_0", "rot_1", "rot_2", "rot_3")
    assert len(names) == 62
    rows = np.stack([el[n] for n in names], axis=1)
    assert np.array_equal(rows, O.ply_rows(m).numpy())
    from luciddreamer_amd import densify as D
    assert tuple(D.ply_attribute_names(15)) == names       # the product's writer uses the reference's property names


def test_build_rotation_matches_reference(ref):
    q = torch.randn(500, 4, generator=torch.Generator().manual_seed(3))
    assert torch.equal(ref.general.build_rotation(q), O.build_rotation(q))
