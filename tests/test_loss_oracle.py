"""The loss oracle (oracle/loss_oracle.py) pinned against outputs of the reference's own functions
(tests/golden/ref_loss_fixtures.npz <- /root/reference/utils/loss.py via tests/golden/make_loss_golden.py)."""
import os

import numpy as np
import pytest

from oracle import loss_oracle

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_loss_fixtures.npz")
CASES = ("small", "tile_edges", "one_channel", "tiny")


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lam", (0.2, 1.0, 0.0))
def test_oracle_matches_reference_loss_and_gradient(case, lam):
    fx = np.load(FIX)
    img, gt = fx[f"{case}_img"], fx[f"{case}_gt"]
    o = loss_oracle.l1_dssim(img, gt, lam)
    # the reference ran in float32; the oracle in float64
    assert abs(o["loss"] - float(fx[f"{case}_lam{lam}_loss"])) <= 2e-6
    ref_g = fx[f"{case}_lam{lam}_grad"]
    assert np.abs(o["grad"] - ref_g).max() <= 2e-5 * max(np.abs(ref_g).max(), 1e-12) + 1e-9


@pytest.mark.parametrize("case", CASES)
def test_oracle_parts(case):
    fx = np.load(FIX)
    o = loss_oracle.l1_dssim(fx[f"{case}_img"], fx[f"{case}_gt"], 0.2, want_grad=False)
    assert abs(o["l1"] - float(fx[f"{case}_l1"])) <= 1e-6
    assert abs(o["ssim"] - float(fx[f"{case}_ssim"])) <= 2e-6


def test_identical_images_give_ssim_one_and_zero_loss():
    rng = np.random.default_rng(0)
    a = rng.random((3, 20, 24))
    o = loss_oracle.l1_dssim(a, a, 0.2)
    assert abs(o["ssim"] - 1.0) < 1e-12 and abs(o["loss"]) < 1e-12
