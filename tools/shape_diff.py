#!/usr/bin/env python
"""Where do the shapes of the blend backward differ?  C3 views through lr_tune_set("blend_quad", 0 / 1 / 2): per view and
tensor, the rows that differ most from shape 0 (diagnostics; the shapes sum the same terms in a different order)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import _lib, cameras, synthetic  # noqa: E402
from tests import helpers as hp  # noqa: E402

dev = torch.device("cuda:0")
P = 1_000_000
cloud = synthetic.make_cloud(P, "band", 0)
path = cameras.rotate360_path(1920, 1080, n_views=30)
g = synthetic.upstream_grad(1080, 1920)
bg = torch.zeros(3)
views = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(0, 30, 3))
for vi in views:
    outs = {}
    for shape in (0, 1):
        _lib.tune_set("blend_quad", shape)
        outs[shape] = hp.run_hip(cloud, path[vi], 3, bg, dev, g)
    line = []
    worst = 0.0
    for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"):
        a, b = outs[1]["grads"][k].reshape(P, -1), outs[0]["grads"][k].reshape(P, -1)
        row = np.abs(a - b).max(axis=1)
        scale = np.abs(b).max()
        i = int(row.argmax())
        worst = max(worst, row.max() / scale)
        line.append(f"{k}: {row.max() / scale:.1e} (row {i}, |row|/max {np.abs(b[i]).max() / scale:.1e}, rows>1e-5: {(row > 1e-5 * scale).sum()})")
    print(f"view {vi}: worst {worst:.1e}  " + "; ".join(line), flush=True)
    if worst > 1e-5:
        i = int(np.abs(outs[1]["grads"]["means2D"] - outs[0]["grads"]["means2D"]).max(axis=1).argmax())
        rad = outs[0]["radii"][i]
        print(f"   Gaussian {i}: radius {rad}; means2D grad shape0 {outs[0]['grads']['means2D'][i]} shape1 {outs[1]['grads']['means2D'][i]}", flush=True)
_lib.tune_set("blend_quad", -1)
