#!/usr/bin/env python
"""Generates tests/golden/ref_raster_fixtures.npz: outputs of oracle/_ref -- the reference's own rasterizer sources
compiled for the host (oracle/build_ref.py) -- on the seeded cases of tests/ref_cases.py.  Needs /root/reference (build
container only).  Images, radii and all nine gradient tensors per case; the reference runs one thread block at a time so
that its float atomicAdds have a fixed order.

    python tests/golden/make_ref_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref                      # noqa: E402
from tests import ref_cases                 # noqa: E402

GRADS = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "conic")
# the small cases; the rest are covered bit for bit wherever oracle/_ref is built
KEEP = ("box_deg3", "near_plane", "fov_clamp", "depth_ties", "sh_clamp", "needles", "cov3D_precomp", "single", "M1")


def main():
    assert ref.available(), "needs /root/reference"
    ref.set_threads(1)
    out = {}
    for case in ref_cases.all_cases():
        n = case["name"]
        if n not in KEEP:
            continue
        r = ref.forward(*ref_cases.forward_args(case))
        g = ref.backward(r, case["dL_dcolor"])
        out[n + "/num_rendered"] = np.int64(r.num_rendered)
        out[n + "/color"], out[n + "/depth"], out[n + "/radii"] = r.color, r.depth, r.radii
        for name, a in zip(GRADS, g):
            out[f"{n}/dL_d{name}"] = a
    path = os.path.join(HERE, "ref_raster_fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", len(out), "arrays")


if __name__ == "__main__":
    main()
