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


@pytest.fixture(autouse=True)
def _restore_knobs():
    yield
    for k in ("preprocess", "bwd_red"):
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
