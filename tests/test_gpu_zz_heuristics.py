"""The kernel shape the library picks by rule (render_bwd.hip blend_shape, render_fwd.hip launch_render_fwd) against every shape
it could have been forced to, on the BASELINE.json workload shapes and on LucidDreamer's own scene statistics (ld512), with three
views in flight and with one: the rule's step against the best forced one (within 3 %).  A reduced form of
tools/shape_sweep.py (profiles/r06q_shape_sweep.json is the full sweep the rule was written from): smaller view counts, the two
kernels varied one at a time.  Measured on this round's boxes: >= 0.99 in every case.  The 3 % target is REPORTED (a warning
below it); the hard assertion is 5 %: two legs of identical kernels differ by up to 2 % on this part, and a timing test must not
be what stops a parity suite (the file name makes it the last module of the GPU run for the same reason)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu

# (workload, views per step): the 3 M / 1440p shape is left to the tool (tens of seconds per sweep)
CASES = [("c3", 9), ("c2", 9), ("c3box", 6), ("c5shape", 9), ("ld512", 12)]


@pytest.mark.parametrize("name,views", CASES)
@pytest.mark.parametrize("in_flight", [3, 1])
def test_the_rule_picks_a_shape_within_3_percent_of_the_best(hip_device, name, views, in_flight):
    import bench
    from luciddreamer_amd import _lib, config
    args = argparse.Namespace(sh_degree=3, no_fused_accumulate=False, gaussians=None, exchange="allreduce")
    wl = bench.Workload(name, args, 0, 1, hip_device, views=views)
    combos = [(-1, -1), (0, -1), (1, -1), (2, -1), (-1, 0), (-1, 2)]          # the rule; each backward shape; each forward shape
    best, shapes = {}, {}
    try:
        for rnd in range(3):
            order = combos[rnd * 2:] + combos[:rnd * 2]                        # no configuration is always the first of a round
            for b, f in order:
                _lib.tune_set("blend_quad", b)
                _lib.tune_set("fwd_pair", f)
                v, _, _, _ = bench.run_leg(wl, "views", False, in_flight, 5, 2, 1, hip_device)
                best[(b, f)] = max(best.get((b, f), 0.0), v)
                shapes[(b, f)] = _lib.last_launch_shapes()
    finally:
        _lib.tune_set("blend_quad", -1)
        _lib.tune_set("fwd_pair", -1)
        config.reset()
        del wl
        torch.cuda.empty_cache()
    # a forced configuration that launched the very kernels the rule picks IS the rule: its measurements count for the rule
    # (two legs of identical kernels differ by up to 2 % on this part: the figure compared must not be noise between the two)
    for k in list(best):
        if k != (-1, -1) and shapes[k] == shapes[(-1, -1)]:
            best[(-1, -1)] = max(best[(-1, -1)], best[k])
    rule = best.pop((-1, -1))
    top = max(best.values())
    print(f"{name} {in_flight} in flight: rule {rule:.1f} views/s, best forced {top:.1f} ({max(best, key=best.get)}), ratio {rule / top:.3f}")
    if rule < 0.97 * top:
        import warnings
        warnings.warn(f"kernel shape rule below the 3 % target: {name}, {in_flight} in flight: {rule:.1f} vs {top:.1f} views/s")
    assert rule >= 0.95 * top, (name, in_flight, rule, best)
