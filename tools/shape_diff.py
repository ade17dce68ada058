#!/usr/bin/env python
"""Where do the shapes of the blend backward differ?  One C3 view through lr_tune_set("blend_quad", 0 / 1 / 2): per tensor,
the rows that differ most from shape 0 (diagnostics; the shapes sum the same terms in a different order)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import _lib, cameras, synthetic  # noqa: E402
from tests import helpers as hp  # noqa: E402

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cloud = synthetic.make_cloud(P, "band", 0)
cam = cameras.rotate360_path(1920, 1080, n_views=30)[0]
g = synthetic.upstream_grad(1080, 1920)
bg = torch.zeros(3)
base = None
for shape in (0, 1, 2, 0, 1):
    _lib.tune_set("blend_quad", shape)
    out = hp.run_hip(cloud, cam, 3, bg, dev, g)["grads"]
    if base is None:
        base = out
        continue
    line = []
    for k in ("means2D", "opacity", "means3D", "sh", "scales", "rotations"):
        a, b = out[k].reshape(P, -1), base[k].reshape(P, -1)
        row = np.abs(a - b).max(axis=1)
        scale = np.abs(b).max()
        i = int(row.argmax())
        line.append(f"{k}: {row.max() / scale:.1e} (row {i}, |row| {np.abs(b[i]).max() / scale:.1e}, rows>1e-5: {(row > 1e-5 * scale).sum()})")
    print(f"shape {shape} vs 0:", "; ".join(line), flush=True)
_lib.tune_set("blend_quad", -1)
