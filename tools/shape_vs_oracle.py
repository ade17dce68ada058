#!/usr/bin/env python
"""Which shape of the blend backward agrees with the CPU oracle on the rows where the shapes disagree?  (diagnostics)"""
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
for vi in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "15,18").split(",")]:
    ref = hp.run_oracle(cloud, path[vi], 3, bg, g)
    st = ref["res"].stage()
    outs = {}
    for shape in (0, 1, 2):
        _lib.tune_set("blend_quad", shape)
        outs[shape] = hp.run_hip(cloud, path[vi], 3, bg, dev, g)
    a, b = outs[0]["grads"]["means2D"], outs[1]["grads"]["means2D"]
    bad = np.nonzero(np.abs(a - b).max(axis=1) > 1e-5 * np.abs(a).max())[0]
    print(f"view {vi}: rows where shapes 0 and 1 differ: {bad.tolist()}")
    for i in bad:
        o = ref["grads"]["means2D"][i]
        m2 = st["means2D"][i]
        fy, fx = np.nonzero(st["fragile"] != 0)
        near = int(((np.abs(fx - m2[0]) < 40) & (np.abs(fy - m2[1]) < 40)).sum())
        print(f"  Gaussian {i} at pixel ({m2[0]:.1f}, {m2[1]:.1f}), radius {ref['radii'][i]}, flagged pixels within 40 px: {near}")
        print(f"     oracle {o[:2]}  shape0 (2 waves) {outs[0]['grads']['means2D'][i][:2]}  shape1 (4 waves) {outs[1]['grads']['means2D'][i][:2]}  shape2 (tile) {outs[2]['grads']['means2D'][i][:2]}")
_lib.tune_set("blend_quad", -1)
