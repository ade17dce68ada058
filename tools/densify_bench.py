#!/usr/bin/env python
"""prune_points on all 21 row tensors of a GaussianModel: luciddreamer_amd.densify (one lr_select_rows call) vs the
reference's way (boolean-mask indexing tensor by tensor + re-created Parameters, gaussian_model.py:273-304)."""
import argparse, json, os, sys, time
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luciddreamer_amd import densify as D      # noqa: E402

ATTR = D.GROUP_ATTR


class Model:
    def __init__(self, P, dev):
        mk = lambda *s: nn.Parameter(torch.randn(*s, device=dev).requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3)
        self._opacity, self._scaling, self._rotation = mk(P, 1), mk(P, 3), mk(P, 4)
        self.percent_dense = 0.01
        self.optimizer = torch.optim.Adam([{"params": [getattr(self, a)], "lr": 1e-3, "name": n} for n, a in ATTR.items()],
                                          lr=0.0, eps=1e-15)
        for a in ATTR.values():
            getattr(self, a).grad = torch.zeros_like(getattr(self, a))
        self.optimizer.step()
        self.xyz_gradient_accum, self.denom = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)


def reference_style_prune(m, mask):
    valid = ~mask
    for group in m.optimizer.param_groups:
        old = group["params"][0]
        st = m.optimizer.state.get(old, None)
        st["exp_avg"] = st["exp_avg"][valid]
        st["exp_avg_sq"] = st["exp_avg_sq"][valid]
        del m.optimizer.state[old]
        group["params"][0] = nn.Parameter(old[valid].requires_grad_(True))
        m.optimizer.state[group["params"][0]] = st
        setattr(m, ATTR[group["name"]], group["params"][0])
    m.xyz_gradient_accum, m.denom, m.max_radii2D = m.xyz_gradient_accum[valid], m.denom[valid], m.max_radii2D[valid]
    torch.cuda.empty_cache()                                   # densify_and_prune ends with this (:403)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    for name, fn in (("lr_select_rows", D.prune_points), ("reference_style", reference_style_prune)):
        m = Model(a.gaussians, dev)
        if name == "lr_select_rows":
            D._store(m)                                         # adoption into the store happens once per model
        ts = []
        for it in range(6):
            mask = torch.rand(m._xyz.shape[0], device=dev) < 0.05
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(m, mask)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[name] = round(min(ts[1:]) * 1e3, 3)
    print(json.dumps({"workload": f"prune 5% of {a.gaussians} Gaussians (6 params + 12 Adam moments + 3 stats)", "ms": res}))


if __name__ == "__main__":
    main()
