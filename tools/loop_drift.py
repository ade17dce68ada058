#!/usr/bin/env python
"""Does the installed unchanged loop slow down over consecutive passes in one process?  Eight passes of 80 iterations (a fresh
model each, as tools/ref_loop_ab.py does), per pass: ms / iteration, the allocator's reserved bytes and hipMalloc count, the
number of objects the Python collector tracks, and the GPU clock rocm-smi reports right after the pass."""
import gc
import os
import subprocess
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


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return " ".join(ln.split(":")[-1].strip() for ln in out.splitlines() if "sclk" in ln)[:40]
    except Exception as e:
        return type(e).__name__


fresh_stack = "--fresh-stack" in sys.argv
for p in range(8):
    config.reset()
    config.set_async(True)
    with ref_loop.stack("ours") as (R, dev):
        h = luciddreamer_amd.install(R)
        try:
            gm = ref_loop.model_from_cloud(R, base, dev)
            cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
            torch.cuda.synchronize()
            if "--freeze" in sys.argv:
                gc.collect()
                gc.freeze()                       # what exists now is never scanned again: a full collection stays cheap
            g0 = [s_["collections"] for s_ in gc.get_stats()]
            pr = None
            if "--profile" in sys.argv and p in (1, 3):
                import cProfile
                pr = cProfile.Profile()
                pr.enable()
            t0 = time.perf_counter()
            ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
            if pr is not None:
                pr.disable()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters * 1e3
            if pr is not None:
                import io
                import pstats
                out = io.StringIO()
                pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(12)
                print("\n".join(ln for ln in out.getvalue().splitlines() if ln.strip())[:2600], flush=True)
        finally:
            luciddreamer_amd.uninstall(h)
    g1 = [s_["collections"] for s_ in gc.get_stats()]
    if "--freeze" in sys.argv:
        gc.unfreeze()
    st = torch.cuda.memory_stats()
    print(f"pass {p}: {dt:.3f} ms / iteration; reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, "
          f"device mallocs {st.get('num_device_alloc', -1)}, frees {st.get('num_device_free', -1)}, gc objects {len(gc.get_objects())}, "
          f"collections in the pass (gen 0/1/2) {[b - a for a, b in zip(g0, g1)]}, sclk {sclk()}", flush=True)
    del gm
