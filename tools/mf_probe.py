#!/usr/bin/env python
"""The MFMA-transposed reduction of the blend backward (lr_tune_set("bwd_red", 5 / 6)) against the swap / DPP reduction
(bwd_red = 1) and against the CPU oracle, per kernel shape (blend_quad 0 / 1 / 2 = 2, 4, 1 waves per tile).  Diagnostics."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import _lib, synthetic  # noqa: E402
from tests import helpers as hp  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["means2D", "opacity", "means3D", "sh", "scales", "rotations"]
worst = 0.0
for (P, W, H, sm) in [(40_000, 640, 360, 1.0), (30_000, 500, 300, 4.0)]:
    cam, cloud = hp.box_setup(P, W, H, seed=3, scale_mult=sm)
    g = synthetic.upstream_grad(H, W)
    bg = torch.tensor([0.1, 0.2, 0.3])
    ref = hp.run_oracle(cloud, cam, 3, bg, g)
    for shape in (1, 0, 2):
        _lib.tune_set("blend_quad", shape)
        outs = {}
        for red in (1, 5, 6):
            _lib.tune_set("bwd_red", red)
            outs[red] = hp.run_hip(cloud, cam, 3, bg, dev, g)["grads"]
        for red in (5, 6):
            line = []
            for k in NAMES:
                scale = float(np.abs(outs[1][k]).max())
                d1 = float(np.abs(outs[red][k] - outs[1][k]).max()) / scale
                do = float(np.abs(outs[red][k] - ref["grads"][k]).max()) / float(np.abs(ref["grads"][k]).max())
                d0 = float(np.abs(outs[1][k] - ref["grads"][k]).max()) / float(np.abs(ref["grads"][k]).max())
                worst = max(worst, d1)
                line.append(f"{k} {d1:.1e} (oracle: {do:.1e} vs {d0:.1e})")
            print(f"P={P} {W}x{H} x{sm} shape {shape} red {red}: " + "; ".join(line), flush=True)
_lib.tune_set("bwd_red", -1)
_lib.tune_set("blend_quad", -1)
print("worst relative difference to the swap / DPP reduction:", worst)
