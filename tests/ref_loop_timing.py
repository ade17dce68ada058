"""Times the unchanged reference training loop (tests/ref_loop.py) at C5's size over this repository's rasterizer and over
the reference's own kernels on the same GPU, after a warm-up pass of each (MIOpen tuning, allocator).  Test infrastructure."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from luciddreamer_amd import cameras          # noqa: E402
from tests import helpers as hp, ref_loop     # noqa: E402
from tests.test_gpu_reference_stack import _perturbed, _targets   # noqa: E402


def main():
    P, W, H, iters = 1_000_000, 512, 512, int(os.environ.get("ITERS", "100"))
    cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
    base, hidden = _perturbed(P, 41)
    targets, depths = _targets(hidden, cams)
    order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=iters)]
    for be in ("ours", "refdev", "ours", "refdev"):
        with ref_loop.stack(be) as (R, dev):
            gm = ref_loop.model_from_cloud(R, base, dev)
            cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
            torch.cuda.synchronize()
            t0 = time.time()
            out = ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
            torch.cuda.synchronize()
            print(f"{be}: {iters} iterations in {time.time() - t0:.2f}s ({(time.time() - t0) / iters * 1e3:.2f} ms/iteration), "
                  f"final loss {out['loss'][-1]:.5f}")


if __name__ == "__main__":
    main()
