"""The three execution shapes of the blend backward kernel (render_bwd.hip: 2 waves per tile with two pixels per lane,
one 8x8 quadrant per wave, or one wave per tile with four pixels per lane) against the CPU oracle on the same scenes.  The library picks the shape from the tile count
(lr_tune_set("blend_quad", .) forces one); each shape runs in its own interpreter with the knob set before anything else."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os
import numpy as np, torch
from luciddreamer_amd import _lib, synthetic
from tests import helpers as hp
_lib.tune_set("blend_quad", int(os.environ["SHAPE_UNDER_TEST"]))
dev = torch.device("cuda:0")
for (P, W, H, seed) in ((20000, 320, 200, 0), (3000, 333, 77, 1), (50000, 640, 360, 2)):
    cam, cloud = hp.box_setup(P, W, H, seed=seed)
    bg = torch.tensor([0.1, 0.2, 0.3])
    g = synthetic.upstream_grad(H, W, seed=seed)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hip = hp.run_hip(cloud, cam, 3, bg, dev, g)
    hp.compare_forward(hip, ref)
    # every row within 1e-4 of its tensor's maximum; rows beyond (at most 8, at most 1e-3) must sit on a pixel the oracle flags
    # as within float32 rounding of a discrete decision (helpers.compare_grads_by_row)
    hp.compare_grads_by_row(hip, ref, P, max_outliers=8)
    assert _lib.last_launch_shapes()[1] == {0: "half", 1: "quad", 2: "tile"}[int(os.environ["SHAPE_UNDER_TEST"])]
print("SHAPE-OK")
"""


@pytest.mark.parametrize("quad", ["0", "1", "2"])
def test_blend_backward_shape_matches_oracle(hip_device, quad):
    env = dict(os.environ, SHAPE_UNDER_TEST=quad, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHAPE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_4k_image_more_tiles_than_partition_bins(hip_device):
    """3840x2160 = 32400 tiles > 16384 partition bins (tilebin.hip): two neighbouring tiles share a bin, the per-bin sort
    orders by (sub-tile, depth, slot) and writes the per-tile ranges itself.  Forward + backward against the oracle, and
    the per-tile lists bit for bit."""
    import numpy as np
    import torch
    from luciddreamer_amd import synthetic
    from tests import helpers as hp
    from tests.test_gpu_full import _raw_forward, _unpack
    W, H, P = 3840, 2160, 60_000
    cam, cloud = hp.box_setup(P, W, H, seed=3, scale_mult=0.7)
    bg = torch.tensor([0.0, 0.1, 0.0])
    g = synthetic.upstream_grad(H, W, seed=3)
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    hip = hp.run_hip(cloud, cam, 3, bg, hip_device, g)
    hp.compare_forward(hip, ref, max_fragile=4e-4 * W * H)     # 8.3 M pixels, splats tens of pixels wide
    hp.compare_grads_by_row(hip, ref, P, max_outliers=16)
    u = _unpack(_raw_forward(cloud, cam, 3, bg, hip_device), P, W, H)
    st = ref["res"].stage()
    rng, orng = u["ranges"].astype(np.int64), st["ranges"].astype(np.int64)
    assert rng.shape[0] == 240 * 135 and int((rng[:, 1] - rng[:, 0]).sum()) == u["point_list"].shape[0]
    for t in np.nonzero(orng[:, 1] > orng[:, 0])[0][::37]:
        ours = u["point_list"][rng[t, 0]:rng[t, 1]]
        theirs = st["point_list"][orng[t, 0]:orng[t, 1]]
        assert np.array_equal(theirs[np.isin(theirs, ours)], ours), t
