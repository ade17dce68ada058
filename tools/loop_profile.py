#!/usr/bin/env python
"""cProfile of the UNCHANGED reference training iteration after luciddreamer_amd.install() (1 M Gaussians, 512 x 512, batch 1):
where the host time of an iteration goes."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import luciddreamer_amd                                  # noqa: E402
from luciddreamer_amd import cameras, config             # noqa: E402
from tests import ref_loop                                # noqa: E402
from tests.test_gpu_reference_stack import _perturbed, _targets   # noqa: E402

P, W, H, iters = 1_000_000, 512, 512, 80
cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
base, hidden = _perturbed(P, 41)
targets, depths = _targets(hidden, cams)
order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=iters)]
config.reset()
config.set_async(True)
with ref_loop.stack("ours") as (R, dev):
    h = luciddreamer_amd.install(R)
    try:
        for rnd in range(2):
            gm = ref_loop.model_from_cloud(R, base, dev)
            cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            if rnd:
                pr.enable()
            ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
            if rnd:
                pr.disable()
            torch.cuda.synchronize()
            print(f"round {rnd}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms / iteration", flush=True)
    finally:
        luciddreamer_amd.uninstall(h)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
from luciddreamer_amd import dropin
print("lazy assignments:", dropin.lazy_assignments)
