"""Not a test: an estimate, from the CPU oracle's per-tile lists of one C3 view, of what a two-stream blend step would save
(DESIGN.md section 4a): the forward walks, per 8x8 quadrant and per chunk of 64 list positions, the candidates that pass the
exact box test of the quadrant (nU steps).  If the upper and the lower 8x4 half of the quadrant each walked their own
candidates side by side (lanes 0-31 / 32-63), a chunk would take max(nT, nB) steps.  Prints sum(nU), sum(max(nT, nB)) and
the one-sided fraction.  Early termination is ignored (both shapes stop at the same position).

    python tests/analysis_half_streams.py [gaussians] [view]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import cameras, synthetic      # noqa: E402
from tests import helpers as hp                      # noqa: E402

CULL_MARGIN = 0.02


def box_hit(mx, my, ca, cb, cc, qmax, x_lo, x_hi, y_lo, y_hi):
    """numpy restatement of common.h box_hit (float64 is fine for a count)."""
    dx_lo, dx_hi = mx - x_hi, mx - x_lo
    dy_lo, dy_hi = my - y_hi, my - y_lo
    inside = (dx_lo <= 0) & (dx_hi >= 0) & (dy_lo <= 0) & (dy_hi >= 0)
    r_c, r_a = -cb / cc, -cb / ca
    qmin = np.full_like(mx, 3e38)
    for ex in (dx_lo, dx_hi):
        ys = np.minimum(dy_hi, np.maximum(dy_lo, r_c * ex))
        qmin = np.minimum(qmin, 0.5 * (ca * ex * ex + cc * ys * ys) + cb * ex * ys)
    for ey in (dy_lo, dy_hi):
        xs = np.minimum(dx_hi, np.maximum(dx_lo, r_a * ey))
        qmin = np.minimum(qmin, 0.5 * (ca * xs * xs + cc * ey * ey) + cb * xs * ey)
    return inside | ~(qmin > qmax)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    W, H = 1920, 1080
    cloud = synthetic.make_cloud(P, "band", 0)
    cam = cameras.rotate360_path(W, H, n_views=30)[view]
    ref = hp.run_oracle(cloud, cam, 3, torch.zeros(3))
    st = ref["res"].stage()
    rng, pl = st["ranges"].astype(np.int64), st["point_list"].astype(np.int64)
    m2, co = st["means2D"].astype(np.float64), st["conic_opacity"].astype(np.float64)
    gx = (W + 15) // 16
    tile_of = np.repeat(np.arange(rng.shape[0]), np.maximum(rng[:, 1] - rng[:, 0], 0))
    pos_in_tile = np.arange(pl.shape[0]) - np.repeat(rng[:, 0], np.maximum(rng[:, 1] - rng[:, 0], 0))
    g = pl
    mx, my, ca, cb, cc, op = m2[g, 0], m2[g, 1], co[g, 0], co[g, 1], co[g, 2], co[g, 3]
    qmax = np.where(op > 0, np.log(255.0 * np.maximum(op, 1e-30)) + CULL_MARGIN, -3e38)
    tx, ty = (tile_of % gx) * 16.0, (tile_of // gx) * 16.0
    chunk = tile_of * 4096 + pos_in_tile // 64                 # (tile, chunk of 64 positions)
    tot_u = tot_m = tot_both = 0
    for q in range(4):
        x0, y0 = tx + (q & 1) * 8, ty + (q >> 1) * 8
        hu = box_hit(mx, my, ca, cb, cc, qmax, x0, x0 + 7, y0, y0 + 7)
        ht = box_hit(mx, my, ca, cb, cc, qmax, x0, x0 + 7, y0, y0 + 3)
        hb = box_hit(mx, my, ca, cb, cc, qmax, x0, x0 + 7, y0 + 4, y0 + 7)
        keys, inv = np.unique(chunk, return_inverse=True)
        nu = np.bincount(inv, weights=hu, minlength=keys.size)
        nt = np.bincount(inv, weights=ht & hu, minlength=keys.size)
        nb = np.bincount(inv, weights=hb & hu, minlength=keys.size)
        tot_u += nu.sum(); tot_m += np.maximum(nt, nb).sum(); tot_both += (ht & hb & hu).sum()
    print(f"instances {pl.shape[0]}, quadrant steps now {int(tot_u)} ({tot_u / pl.shape[0]:.2f} per instance), "
          f"two-stream steps {int(tot_m)} ({tot_m / tot_u:.3f} of now), candidates touching both halves {tot_both / tot_u:.3f}")


if __name__ == "__main__":
    main()
