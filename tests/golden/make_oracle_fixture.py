#!/usr/bin/env python
"""Generates tests/golden/oracle_e2e_fixture.npz: seeded inputs are regenerated from
luciddreamer_amd.synthetic (seed in the file), expected outputs come from the CPU oracle.

The reference ships no golden vectors for the rasterizer, so this end-to-end
fixture pins OUR oracle against drift and lets the GPU tests compare without rebuilding the oracle.
    python tests/golden/make_oracle_fixture.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from luciddreamer_amd import synthetic  # noqa: E402
from tests import helpers as hp  # noqa: E402

SPEC = dict(P=2500, W=112, H=80, seed=7, degree=3, bg=(0.2, 0.1, 0.4), grad_seed=11)


def main():
    cam, cloud = hp.box_setup(SPEC["P"], SPEC["W"], SPEC["H"], seed=SPEC["seed"], scale_mult=1.5)
    bg = torch.tensor(SPEC["bg"])
    g = synthetic.upstream_grad(SPEC["H"], SPEC["W"], seed=SPEC["grad_seed"])
    ref = hp.run_oracle(cloud, cam, SPEC["degree"], bg, g)
    st = ref["res"].stage()
    out = dict(color=ref["color"], depth=ref["depth"], radii=ref["radii"], num_rendered=np.int64(ref["num_rendered"]),
               fragile=st["fragile"], n_contrib=st["n_contrib"], final_T=st["final_T"], ranges=st["ranges"],
               point_list=st["point_list"])
    for k, v in ref["grads"].items():
        out["grad_" + k] = v
    path = os.path.join(HERE, "oracle_e2e_fixture.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
