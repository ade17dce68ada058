#!/usr/bin/env python
"""Generates tests/golden/ref_loss_fixtures.npz by IMPORTING the reference's own loss functions
(/root/reference/utils/loss.py: l1_loss :18-19, ssim :37-69) and composing them exactly as the training loop does
(/root/reference/luciddreamer.py:301-304), on CPU.  Only the build container can run this; the fixtures travel.

`utils/loss.py` imports cv2 at module level for an unrelated helper (image2canny) and moves a 3x3 helper convolution
to "cuda" at import time (:80-87, nearMean_map); neither exists here, so an empty stand-in module is registered for
cv2 and nn.Module.cuda is a no-op during the import -- none of the functions used below touch either.

    python tests/golden/make_loss_golden.py
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    with mock.patch.object(torch.nn.Module, "cuda", lambda self, device=None: self):
        from utils.loss import l1_loss, ssim                  # reference code

    out = {}
    g = torch.Generator().manual_seed(77)
    cases = {"small": (3, 37, 45), "tile_edges": (3, 64, 96), "one_channel": (1, 33, 31), "tiny": (3, 7, 9)}
    for name, (C, H, W) in cases.items():
        gt = torch.rand(C, H, W, generator=g)
        # a rendered image that resembles the target: blurred target + noise, clamped like a render
        img = (0.7 * gt + 0.3 * torch.rand(C, H, W, generator=g)).clamp(0, 1)
        img[:, : H // 3] = gt[:, : H // 3]                    # a region of exact agreement (|d| = 0: sign(0) = 0)
        for lam in (0.2, 1.0, 0.0):
            x = img.clone().requires_grad_(True)
            Ll1 = l1_loss(x, gt)                                               # luciddreamer.py:302
            s = ssim(x, gt)
            loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)                         # luciddreamer.py:303
            loss.backward()
            out[f"{name}_lam{lam}_loss"] = np.float32(loss.item())
            out[f"{name}_lam{lam}_grad"] = x.grad.numpy().copy()
        out[f"{name}_img"], out[f"{name}_gt"] = img.numpy(), gt.numpy()
        out[f"{name}_l1"], out[f"{name}_ssim"] = np.float32(l1_loss(img, gt).item()), np.float32(ssim(img, gt).item())
    path = os.path.join(HERE, "ref_loss_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]})


if __name__ == "__main__":
    main()
