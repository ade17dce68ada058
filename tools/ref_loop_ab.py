#!/usr/bin/env python
"""The unchanged reference training iteration (tests/ref_loop.py: R/luciddreamer.py:283-327 around the reference's own
GaussianModel / render() / loss) at 1 M Gaussians, 512 x 512, batch 1, over this rasterizer under the three host-sync modes
of luciddreamer_amd.config, measured alternately in one process: exact (the reference's read-back in the middle of every
forward), verify (default: async, the forward waits for its own early header), drop (async, nothing waits)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from luciddreamer_amd import cameras, config          # noqa: E402
from tests import ref_loop                               # noqa: E402
from tests.test_gpu_reference_stack import _perturbed, _targets   # noqa: E402


def main():
    P, W, H, iters = 1_000_000, 512, 512, int(os.environ.get("ITERS", "80"))
    cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
    base, hidden = _perturbed(P, 41)
    targets, depths = _targets(hidden, cams)
    order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=iters)]
    import luciddreamer_amd
    modes = {"exact": lambda: config.set_async(False), "verify": lambda: config.set_async(True),
             "drop": lambda: config.set_async(True, on_overflow="drop")}
    if "--install" in sys.argv:                              # the same loop after luciddreamer_amd.install(R)
        modes = {k + "+install": v for k, v in modes.items()}
    res = {k: [] for k in modes}
    for rnd in range(3):
        for name, setup in modes.items():
            config.reset()
            setup()
            with ref_loop.stack("ours") as (R, dev):
                handle = luciddreamer_amd.install(R, lazy_filter=not os.environ.get("LR_NO_LAZY_FILTER")) if name.endswith("+install") else None
                try:
                    gm = ref_loop.model_from_cloud(R, base, dev)
                    cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                finally:
                    if handle is not None:
                        luciddreamer_amd.uninstall(handle)
            config.drain()
            if rnd:                                         # round 0 warms up (MIOpen, allocator)
                res[name].append(dt / iters * 1e3)
            print(f"round {rnd} {name}: {dt / iters * 1e3:.3f} ms/iteration, final loss {out['loss'][-1]:.5f}", flush=True)
    config.set_async(True)
    print({k: [round(x, 3) for x in v] for k, v in res.items()})


if __name__ == "__main__":
    main()
