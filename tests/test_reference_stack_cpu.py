"""CPU: the reference's own Python layer (render(), GaussianModel, autograd op, l1/ssim) driven end to end over the two
CPU backends -- the compiled reference sources (oracle/_ref) and the restatement -- through tests/ref_loop.py.  Proves
that the harness used by the GPU parity tests runs the reference's classes unchanged, and that an optimisation driven
through the restatement follows the one driven through the reference's own rasterizer code."""
import numpy as np
import pytest
import torch

from luciddreamer_amd import cameras, synthetic
from oracle import ref, ref_python as rp
from tests import helpers as hp, ref_loop

pytestmark = pytest.mark.skipif(not (rp.available() and ref.available()), reason="reference sources not present")


def _setup(P=1500, W=96, H=64):
    cams = cameras.lookaround_path(W, H, n_views=3, max_yaw_deg=10.0, max_pitch_deg=5.0)
    base = synthetic.make_cloud(P, "box", 21, scale_mult=2.0)
    hidden = synthetic.make_cloud(P, "box", 21, scale_mult=2.0)
    hidden["means3D"] = hidden["means3D"] + 0.03 * torch.randn(P, 3, generator=torch.Generator().manual_seed(4))
    hidden["shs"][:, 0] += 0.3
    outs = [hp.run_oracle(hidden, c, 3, torch.zeros(3)) for c in cams]
    targets = [torch.from_numpy(o["color"]) for o in outs]
    depths = [torch.from_numpy(o["depth"]) for o in outs]
    return cams, base, targets, depths


def _run(backend, iters, **kw):
    cams, base, targets, depths = _setup()
    with ref_loop.stack(backend) as (R, dev):
        gm = ref_loop.model_from_cloud(R, base, dev)
        out = ref_loop.train(R, gm, dev, cams, [i % 3 for i in range(iters)], targets, depths, iters=iters, **kw)
        out["xyz"] = gm.get_xyz.detach().clone()
    return out


def test_loop_over_restatement_follows_loop_over_compiled_reference():
    a = _run("ref", 12)
    b = _run("port", 12)
    assert a["loss"][-1] < a["loss"][0]
    assert np.abs(a["loss"] - b["loss"]).max() <= 1e-6 * a["loss"].max()
    assert torch.allclose(a["xyz"], b["xyz"], atol=1e-6)


def test_densification_inside_the_loop_changes_P_identically():
    a = _run("ref", 9, densify_from=2, densify_every=3)
    b = _run("port", 9, densify_from=2, densify_every=3)
    assert len(set(a["P"].tolist())) > 1
    assert np.array_equal(a["P"], b["P"])
    assert np.abs(a["loss"] - b["loss"]).max() <= 1e-5 * a["loss"].max()


def test_create_from_pcd_uses_the_knn_backend():
    rng = np.random.default_rng(0)
    pts = rng.uniform((-1, -1, 2), (1, 1, 4), size=(400, 3)).astype(np.float32)
    cols = rng.uniform(size=(400, 3)).astype(np.float32)
    scal = {}
    for be in ("ref", "port"):
        with ref_loop.stack(be) as (R, dev):
            gm = R.gaussian_model.GaussianModel(3)
            gm.create_from_pcd(rp.PointCloud(pts, cols), 1.0)        # R/scene/gaussian_model.py:126-149 -> distCUDA2
            scal[be] = gm._scaling.detach().clone()
            assert gm._features_rest.shape == (400, 15, 3) and float(gm.get_opacity.mean()) == pytest.approx(0.1, abs=1e-6)
    assert torch.equal(scal["ref"], scal["port"])
