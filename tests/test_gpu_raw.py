"""raw-parameter fast path (lr_forward_raw / lr_backward_raw).

The stored GaussianModel tensors (log-scales, logit-opacities, unnormalised quaternions, SH split in
features_dc | features_rest; /root/reference/scene/gaussian_model.py:47-52, 97-117) go straight into the
kernels.  Checked against
  (a) the standard HIP op fed with torch's own exp / normalize / sigmoid / cat (forward, and backward through
      torch autograd of those activations), and
  (b) the CPU oracle fed with the same activated inputs.
Tolerances as in helpers.py; the activation functions themselves differ from torch's by <= 1-2 ulp.
"""
import math

import numpy as np
import pytest
import torch

from tests import helpers as h
from luciddreamer_amd import cameras, synthetic

pytestmark = pytest.mark.gpu


def _pc(cloud, device, degree, drop_rest=False):
    from luciddreamer_amd.gaussian_renderer import GaussianCloud
    shs = cloud["shs"][:, :1, :] if drop_rest else cloud["shs"]
    c = {k: v.to(device) for k, v in cloud.items()}
    # un-normalise the quaternions so that the normalisation Jacobian is exercised
    g = torch.Generator().manual_seed(5)
    rot = c["rotations"] * (0.5 + 1.5 * torch.rand(c["rotations"].shape[0], 1, generator=g).to(device))
    return GaussianCloud(c["means3D"], c["scales"], rot, c["opacities"], shs.to(device), active_sh_degree=degree)


def _grads(pc):
    return {n: (getattr(pc, n).grad.detach().cpu().numpy() if getattr(pc, n).grad is not None else None)
            for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}


def _zero(pc):
    for p in pc.parameters():
        p.grad = None


@pytest.mark.parametrize("degree,drop_rest", [(3, False), (2, False), (1, False), (0, False), (0, True)])
def test_raw_matches_activated_path(hip_device, degree, drop_rest):
    device = hip_device
    from luciddreamer_amd.gaussian_renderer import render, render_raw
    W, H, P = 256, 192, 6000
    cam, cloud = h.box_setup(P, W, H, seed=11)
    cam = cam.to(hip_device)
    pc = _pc(cloud, device, degree, drop_rest)
    bg = torch.tensor([0.2, 0.1, 0.3], device=device)
    gcol = synthetic.upstream_grad(H, W, seed=3).to(hip_device)

    out_a = render(cam, pc, bg_color=bg)
    (out_a["render"] * gcol).sum().backward()
    g_a, vs_a = _grads(pc), out_a["viewspace_points"].grad.detach().cpu().numpy()
    _zero(pc)
    out_r = render_raw(cam, pc, bg_color=bg)
    (out_r["render"] * gcol).sum().backward()
    g_r, vs_r = _grads(pc), out_r["viewspace_points"].grad.detach().cpu().numpy()

    # forward: the same kernels see inputs that differ by the rounding of the activations only
    same_radii = (out_a["radii"] == out_r["radii"]).float().mean().item()
    assert same_radii >= 0.999, same_radii
    cerr = (out_a["render"] - out_r["render"]).abs().max().item()
    derr = (out_a["depth"] - out_r["depth"]).abs().max().item()
    assert cerr <= 2e-5 and derr <= 2e-4, (cerr, derr)
    assert torch.equal(out_a["visibility_filter"], out_r["visibility_filter"]) or same_radii >= 0.999

    # backward: gradients w.r.t. the STORED tensors vs torch autograd through exp/normalize/sigmoid/cat
    h.compare_grads({"vs": vs_r}, {"vs": vs_a}, names=("vs",), rtol=2e-4)
    for n in g_a:
        if g_a[n] is None or g_a[n].size == 0:
            assert g_r[n] is None or g_r[n].size == 0 or not np.any(g_r[n])
            continue
        h.compare_grads({n: g_r[n]}, {n: g_a[n]}, names=(n,), rtol=2e-4)
    if not drop_rest and degree < 3:
        # bands above the active degree receive exactly zero gradient
        K = (degree + 1) ** 2
        assert not np.any(g_r["_features_rest"][:, K - 1:, :])


def test_raw_matches_cpu_oracle(hip_device):
    device = hip_device
    from luciddreamer_amd.gaussian_renderer import render_raw
    W, H, P = 320, 240, 8000
    cam, cloud = h.box_setup(P, W, H, seed=2)
    pc = _pc(cloud, device, 3)
    bg = torch.tensor([0.0, 0.0, 0.0])
    gcol = synthetic.upstream_grad(H, W, seed=4)
    out = render_raw(cam.to(hip_device), pc, bg_color=bg.to(hip_device))
    (out["render"] * gcol.to(hip_device)).sum().backward()

    # the oracle gets the activated inputs (numpy float32 activations of the same stored tensors)
    act = dict(means3D=pc._xyz.detach().cpu(), shs=pc.get_features.detach().cpu(),
               opacities=pc.get_opacity.detach().cpu(), scales=pc.get_scaling.detach().cpu(),
               rotations=pc.get_rotation.detach().cpu())
    ref = h.run_oracle(act, cam, 3, bg, grad_color=gcol)
    hip = dict(color=out["render"].detach().cpu().numpy(), depth=out["depth"].detach().cpu().numpy(),
               radii=out["radii"].cpu().numpy())
    same = (hip["radii"] == ref["radii"]).mean()
    assert same >= 0.999, same                    # activations are not bit-identical to torch's
    h.compare_forward(hip, ref, check_exact=False)
    # chain the oracle's activated-input gradients through the activations in float64
    s = pc.get_scaling.detach().cpu().double().numpy()
    o = pc.get_opacity.detach().cpu().double().numpy()
    r = pc._rotation.detach().cpu().double().numpy()
    nr = np.linalg.norm(r, axis=1, keepdims=True)
    q = r / nr
    gq = ref["grads"]["rotations"].astype(np.float64)
    exp = {
        "_xyz": ref["grads"]["means3D"],
        "_scaling": ref["grads"]["scales"] * s,
        "_opacity": ref["grads"]["opacity"].reshape(-1, 1) * o * (1 - o),
        "_rotation": (gq - q * (q * gq).sum(1, keepdims=True)) / nr,
        "_features_dc": ref["grads"]["sh"][:, :1, :],
        "_features_rest": ref["grads"]["sh"][:, 1:, :],
    }
    got = _grads(pc)
    for n, e in exp.items():
        h.compare_grads({n: got[n]}, {n: e.astype(np.float32)}, names=(n,))


def test_raw_fused_accumulation_over_views(hip_device):
    """Two views accumulated by the kernels into existing .grad == sum of the per-view gradients."""
    device = hip_device
    from luciddreamer_amd import config
    from luciddreamer_amd.gaussian_renderer import render_raw
    W, H, P = 256, 192, 5000
    cloud = synthetic.make_cloud(P, "band", 1)
    path = cameras.rotate360_path(W, H, n_views=2)
    cams = [c.to(hip_device) for c in path]
    pc = _pc(cloud, device, 3)
    gcol = synthetic.upstream_grad(H, W, seed=6).to(hip_device)

    per_view = []
    for c in cams:
        _zero(pc)
        (render_raw(c, pc)["render"] * gcol).sum().backward()
        per_view.append(_grads(pc))
    _zero(pc)
    config.set_fused_grad_accumulation(True)
    try:
        for p in pc.parameters():
            p.grad = torch.zeros_like(p)
        for c in cams:
            (render_raw(c, pc)["render"] * gcol).sum().backward()
        got = _grads(pc)
    finally:
        config.set_fused_grad_accumulation(False)
    for n in got:
        want = per_view[0][n] + per_view[1][n]
        h.compare_grads({n: got[n]}, {n: want}, names=(n,), rtol=1e-5)


def test_raw_rejects_missing_inputs(hip_device):
    device = hip_device
    from luciddreamer_amd import _C
    cam = cameras.identity_camera(64, 64).to(hip_device)
    z = lambda *s: torch.zeros(*s, device=device)
    with pytest.raises(RuntimeError):
        _C.rasterize_gaussians_raw(z(3), z(4, 3), z(4, 1, 3), z(4, 15, 3), None, z(4, 3), z(4, 4), 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 0.4, 0.4, 64, 64, 3,
                                   cam.camera_center, False)


def test_raw_path_under_view_streams_with_grad_output(hip_device):
    """ViewStreams.run_view(..., grad_output=g) with the default direct=True used to CALL the autograd node
    of the view's output -- only the compiled node of the standard operator is callable; render_raw's node belongs to a Python
    autograd.Function ('...Backward' object is not callable).  Such views must go through the engine, and give the gradients
    of the per-view backward."""
    from luciddreamer_amd import config, parallel
    from luciddreamer_amd.gaussian_renderer import render_raw
    W, H, P = 256, 192, 6000
    cam, cloud = h.box_setup(P, W, H, seed=11)
    cams = [cam.to(hip_device)] * 4
    bg = torch.tensor([0.2, 0.1, 0.3], device=hip_device)
    gcol = synthetic.upstream_grad(H, W, seed=3).to(hip_device)

    def run(piped):
        pc = _pc(cloud, hip_device, 3)
        config.set_fused_grad_accumulation(True)
        try:
            if piped:
                pipe = parallel.ViewStreams(hip_device, 2)
                pipe.begin_step()
                for c in cams:
                    pipe.run_view(lambda c=c: render_raw(c, pc, bg_color=bg)["render"], grad_output=gcol)
                pipe.end_step()
            else:
                for c in cams:
                    render_raw(c, pc, bg_color=bg)["render"].backward(gcol)
            torch.cuda.synchronize()
        finally:
            config.set_fused_grad_accumulation(False)
        return _grads(pc)

    a, b = run(False), run(True)
    for n in a:
        assert a[n] is not None and np.abs(a[n]).max() > 0, n
        assert np.abs(a[n] - b[n]).max() <= 1e-5 * np.abs(a[n]).max(), n
