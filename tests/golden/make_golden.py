#!/usr/bin/env python
"""Generates tests/golden/ref_python_fixtures.npz by IMPORTING the reference's own Python code from
/root/reference (available only in the build container; the GPU box never runs this).

These are the only pieces of the hot path for which the reference ships an executable restatement
(SURVEY.md section 4 / 8c):
  * SH -> RGB:  utils/sh.py:57-112 eval_sh, used as in gaussian_renderer/__init__.py:73-78
                (colors = clamp_min(eval_sh(...) + 0.5, 0))
  * scale/rotation -> covariance: scene/gaussian_model.py:29-33 via utils/general.py:67-116
  * camera matrices: utils/graphics.py:41-75 (getWorld2View2, getProjectionMatrix), the transposes of
    scene/cameras.py:58-61.
The oracle (oracle/raster_oracle.c) and luciddreamer_amd.cameras are pinned against these vectors by
tests/test_oracle_golden.py.

    python tests/golden/make_golden.py
"""
import math
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from utils.sh import eval_sh                       # reference code
    from utils import general as ref_general           # reference code (hard-codes device="cuda")
    from utils.graphics import getWorld2View2, getProjectionMatrix   # reference code

    g = torch.Generator().manual_seed(1234)
    P = 96
    out = {}

    # ---- Gaussians in front of an identity camera so that every one is visible ----
    means = torch.rand(P, 3, generator=g) * torch.tensor([1.6, 0.9, 2.0]) + torch.tensor([-0.8, -0.45, 3.0])
    campos = torch.tensor([0.1, -0.2, 0.05])
    sh = torch.randn(P, 16, 3, generator=g) * 0.4
    sh[:, 0] += 0.6
    out["means"], out["campos"], out["sh"] = means.numpy(), campos.numpy(), sh.numpy()
    dirs = means - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    for deg in range(4):
        shs_view = sh.transpose(1, 2).view(-1, 3, 16)                 # gaussian_renderer/__init__.py:74
        sh2rgb = eval_sh(deg, shs_view, dirs)                         # :77
        out[f"rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()   # :78

    # ---- covariance from scaling / rotation ----
    scaling = torch.exp(torch.randn(P, 3, generator=g) * 0.5 - 3.0)
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)          # build_rotation normalises; the CUDA kernel does not
    out["scaling"], out["rotation"] = scaling.numpy(), rot.numpy()
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    for mod in (1.0, 1.7):
        with mock.patch.object(torch, "zeros", cpu_zeros):
            L = ref_general.build_scaling_rotation(mod * scaling, rot)     # scene/gaussian_model.py:30
            cov = L @ L.transpose(1, 2)                                    # :31
            symm = ref_general.strip_symmetric(cov)                        # :32
        out[f"cov3D_mod{mod}"] = symm.numpy()

    # ---- camera matrices ----
    th = 0.3
    R = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    t = np.array([0.2, -0.1, 0.4])
    out["cam_R"], out["cam_t"] = R, t
    out["world_view_transform"] = torch.tensor(getWorld2View2(R, t)).transpose(0, 1).numpy()   # scene/cameras.py:58
    fovx, fovy = 0.8279103882874479, 0.5
    out["fov"] = np.array([fovx, fovy])
    out["projection_matrix"] = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1).numpy()

    path = os.path.join(HERE, "ref_python_fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
