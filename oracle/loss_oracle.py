"""CPU oracle of the training loop's photometric loss (TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product path).

Restates, in float64 torch on the CPU,
    l1_loss      /root/reference/utils/loss.py:18-19     mean |x - gt|
    gaussian     :26-28                                  11 taps, sigma 1.5, normalised (values formed in float32)
    create_window:31-35                                  outer product, one copy per channel (grouped conv)
    _ssim        :48-69                                  five zero-padded (5) window means, C1 = 0.01^2, C2 = 0.03^2,
                                                         SSIM map, mean over everything
composed as /root/reference/luciddreamer.py:301-304:  (1 - lam) * L1 + lam * (1 - ssim).
The gradient comes from autograd of this restatement.  Pinned against outputs of the reference functions
themselves (tests/golden/ref_loss_fixtures.npz, made by tests/golden/make_loss_golden.py) in tests/test_loss_oracle.py.
"""
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


def window_1d():
    g = torch.tensor([exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)   # :27
    return g / g.sum()                                                                                           # :28


def ssim_map(x, gt):
    """x, gt: [C,H,W] float64 -> SSIM map [C,H,W] (loss.py:48-64)."""
    C = x.shape[0]
    w1 = window_1d().unsqueeze(1)
    w2 = (w1 @ w1.t()).float().to(x.dtype)                                   # :33 (2-D window formed in float32)
    win = w2.expand(C, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=C)[0]
    mu1, mu2 = conv(x), conv(gt)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(x * x) - mu1_sq
    s2 = conv(gt * gt) - mu2_sq
    s12 = conv(x * gt) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def l1_dssim(image, gt, lam, want_grad=True):
    """numpy/torch [C,H,W] -> dict(loss, l1, ssim, grad) in float64."""
    x = torch.as_tensor(np.asarray(image), dtype=torch.float64).clone().requires_grad_(want_grad)
    g = torch.as_tensor(np.asarray(gt), dtype=torch.float64)
    l1 = (x - g).abs().mean()
    s = ssim_map(x, g).mean()
    loss = (1.0 - lam) * l1 + lam * (1.0 - s)
    out = dict(loss=float(loss.detach()), l1=float(l1.detach()), ssim=float(s.detach()))
    if want_grad:
        loss.backward()
        out["grad"] = x.grad.numpy()
    return out
