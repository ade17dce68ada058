#!/usr/bin/env python
"""bench.py -- views/s forward+backward of the MI355X Gaussian-splat rasterizer (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c3box|c4shape|c5shape] [--views V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic views: every rank renders its V views of the
camera path forward+backward (upstream gradient dL/dcolor = N(0,1)), gradients accumulate in one flat fp32 bucket,
and (N > 1) the bucket is all-reduced over RCCL (per parameter tensor, asynchronously, joined before the step ends).
Weak scaling: V views per rank per step, N*V distinct views per step.  Inputs are resident in HBM before the timed
region; data is synthetic.

Workloads (BASELINE.json configs):
  c3 (default; the metric's configuration): 1e6 Gaussians, SH degree 3, 1920x1080, band cloud, rotate360 camera path,
      V = 30 views per rank per step.
  c2: 1e5 Gaussians, box cloud, identity view.  c3box: the dense reading of C3 (all 1e6 Gaussians in view).
  c4shape: 3e6 Gaussians at 2560x1440, all in view.  c5shape: 1e6 Gaussians at 512x512, all in view.

Prints ONE JSON line (rank 0).  `value` is the headline entry point (one lr_views_accumulate call per step, async mode,
3 streams).  In the default run (N = 1, workload c3) the same process also times, with the same K and W,
  * `entry_points`: the drop-in autograd operator (`drop_in_views_per_s`), the same in the reference's exact mode with a
    host round trip per view (`exact_mode_views_per_s`) and the views step with the fused L1+DSSIM loss inside
    (`views_loss_views_per_s`);
  * `other_workloads`: the dense C3-box and C4-shape lines (their 20 %-of-roofline figures, driver-timed).
`roofline` describes the dominant kernel (HIP-event timed inside the library on the launch stream, single stream);
`cpu_baseline` is the CPU oracle ("port" of the reference semantics) on this box's host cores, one view at a time.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
NCOEF = {0: 1, 1: 4, 2: 9, 3: 16}
WORKLOADS = {
    # name: (cloud kind, Gaussians, resolution, views per rank per step, tag)
    "c3": ("band", 1_000_000, (1920, 1080), 30, "C3"),
    "c2": ("box", 100_000, (1920, 1080), 30, "C2"),
    "c3box": ("box", 1_000_000, (1920, 1080), 30, "C3-box (all Gaussians in front of the camera)"),
    "c4shape": ("box", 3_000_000, (2560, 1440), 6, "C4-shape (all Gaussians in front of the camera)"),
    "c5shape": ("box", 1_000_000, (512, 512), 30, "C5-shape (all Gaussians in front of the camera)"),
}


def stage_bytes(stage, P, V, R, N, T, K, M):
    """Algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 3, one read + one write per stage)."""
    if stage == "preprocess":
        return 12 * P + V * (32 + 12 * K) + 8 * P + 40 * V
    if stage == "render_fwd":
        return 44 * R + 24 * N
    if stage == "render_bwd":
        return 40 * R + 20 * N + 44 * V
    if stage == "gauss_bwd":
        return P * (108 + 12 * M) + 92 * V + V * (135 + 24 * K)
    return None


def path_bytes(P, V, R, N, T, K, M):
    b_f = (12 * P + V * (32 + 12 * K)) + (8 * P + 40 * V) + 8 * P + (20 * V + 12 * R) + 24 * R + (8 * R + 8 * T) + (44 * R + 24 * N)
    b_b = (40 * R + 20 * N + 44 * V) + P * (108 + 12 * M) + 92 * V + V * (135 + 24 * K)
    return b_f, b_b


def moved_bytes(P, V, R, N, T, K, M, views_per_step):
    """The section-8d model minus what this design never moves: the reference zero-fills P(108+12M) bytes of gradients
    per backward (RAST/rasterize_points.cu:154-162); here culled rows are not touched and the bucket is zeroed once
    per STEP (59 floats per Gaussian), i.e. 236 P / views_per_step bytes per view."""
    b_f, b_b = path_bytes(P, V, R, N, T, K, M)
    return b_f + b_b - P * (108 + 12 * M) + 236.0 * P / max(views_per_step, 1)


class Workload:
    """Device-resident inputs of one workload and the step functions of every entry point."""

    def __init__(self, name, args, rank, world, dev, gaussians=None, views=None, resolution=None):
        from luciddreamer_amd import _C, cameras, parallel, synthetic
        from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        kind, P, (W, H), V, tag = WORKLOADS[name]
        P, V = gaussians or P, views or V
        if resolution:
            W, H = resolution
        self.name, self.P, self.W, self.H, self.V, self.dev, self.world = name, P, W, H, V, dev, world
        self.degree = args.sh_degree
        cloud = synthetic.make_cloud(P, kind, 0)
        if kind == "band":
            path = cameras.rotate360_path(W, H, n_views=V * world)
            my_cams = [path[i] for i in parallel.shard_views(len(path), rank, world)]
            self.label = f"{tag}: {P} Gaussians, SH degree {self.degree}, {W}x{H}, band cloud, rotate360 path, {V} views/rank/step"
        else:
            my_cams = [cameras.identity_camera(W, H)] * V
            self.label = f"{tag}: {P} Gaussians, SH degree {self.degree}, {W}x{H}, box cloud, identity view x{V}/rank/step"
        self.cloud, self.my_cams = cloud, my_cams
        self.M = cloud["shs"].shape[1]
        self.K, self.N = NCOEF[self.degree], W * H
        self.T = ((W + 15) // 16) * ((H + 15) // 16)
        self.leaf = {k: v.to(dev).requires_grad_(True) for k, v in cloud.items()}
        leaf = self.leaf
        self.params = [leaf["means3D"], leaf["scales"], leaf["rotations"], leaf["opacities"], leaf["shs"]]
        self.grads = parallel.FlatGrads(self.params)
        self.means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
        self.m2d_grad = torch.zeros(P, 3, device=dev)
        self.grad_color = synthetic.upstream_grad(H, W).to(dev)
        self.bg = torch.zeros(3, device=dev)
        self.cams = [c.to(dev) for c in my_cams]
        self.rasterizers = []
        for c in self.cams:
            rs = GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), self.bg, 1.0,
                                               c.world_view_transform, c.full_proj_transform, self.degree, c.camera_center,
                                               False, False)
            self.rasterizers.append(GaussianRasterizer(rs))
        # per-view R (= num_rendered) and V (= visible count), measured once in exact mode; also seeds the async-mode
        # binning capacity so that no view in the timed region needs a host round trip
        empty = torch.Tensor([])
        stats, seen = [], {}
        with torch.no_grad():
            for c, r in zip(self.cams, self.rasterizers):
                key = id(c.world_view_transform)
                if key not in seen:
                    rs = r.raster_settings
                    out = _C.rasterize_gaussians(self.bg, leaf["means3D"], empty, leaf["opacities"], leaf["scales"],
                                                 leaf["rotations"], 1.0, empty, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                                 rs.tanfovy, H, W, leaf["shs"], self.degree, rs.campos, False, False)
                    seen[key] = (int(out[0]), int((out[3] > 0).sum().item()))
                stats.append(seen[key])
        self.view_stats = stats
        self.R_mean = sum(s[0] for s in stats) / len(stats)
        self.V_mean = sum(s[1] for s in stats) / len(stats)
        self.capacity = int(max(s[0] for s in stats) * 1.25) + 4096
        self.batch = None

    def make_step(self, api, exact, streams, fused=True):
        """Returns step(): one optimisation step's worth of views through the given entry point."""
        from luciddreamer_amd import config, parallel
        leaf, dev = self.leaf, self.dev
        config.reset()
        config.set_fused_grad_accumulation(fused)
        config.set_async(False)
        if not exact:
            config.set_async(True, headroom=1.25)
            config._hwm[(dev.index, self.P, self.H, self.W)] = max(s[0] for s in self.view_stats)
        self.batch = None
        if api in ("views", "views-loss"):
            named = {"means3D": leaf["means3D"], "scales": leaf["scales"], "rotations": leaf["rotations"],
                     "opacity": leaf["opacities"], "sh": leaf["shs"]}
            if api == "views":
                self.batch = parallel.ChunkedViewStep(self.cams, [self.grad_color] * len(self.cams), named, self.degree,
                                                      self.bg, self.capacity, n_streams=streams)
            else:
                target = torch.rand(3, self.H, self.W, generator=torch.Generator().manual_seed(2)).to(dev)
                self.batch = parallel.ChunkedViewStep(self.cams, None, named, self.degree, self.bg, self.capacity,
                                                      n_streams=streams, targets=[target] * len(self.cams))
            self.grads = self.batch.grads                   # the parameters' .grad now live in this step's bucket
            batch = self.batch

            def step():
                self.m2d_grad.zero_()
                batch.run(self.m2d_grad)
            return step
        grads = self.grads = parallel.FlatGrads(self.params)
        pipe = parallel.ViewStreams(dev, streams)
        means2D, grad_color = self.means2D, self.grad_color

        def step():
            grads.zero_()
            means2D.grad = self.m2d_grad.zero_() if fused else None
            pipe.begin_step()
            for r in self.rasterizers:
                pipe.run_view(
                    lambda r=r: r(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"],
                                  shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])[0],
                    lambda color: color.backward(grad_color))
            pipe.end_step()
            grads.all_reduce()
        return step

    def finish(self):
        from luciddreamer_amd import config
        config.drain()
        if self.batch is not None:
            self.batch.check()


def timed(step, n_steps, world, dev):
    """K steps between barrier + synchronize on both sides; MAX over ranks.  Returns (seconds, host ms to issue a step)."""
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    host_issue_ms = (time.perf_counter() - t0) / n_steps * 1e3       # host time to ISSUE a step (no device sync yet)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, host_issue_ms


def run_leg(wl, api, exact, streams, steps, warmup, world, dev, fused=True):
    step = wl.make_step(api, exact, streams, fused)
    for _ in range(warmup):
        step()
    dt, host_ms = timed(step, steps, world, dev)
    wl.finish()
    return world * wl.V * steps / dt, dt / steps * 1e3, host_ms, step


def roofline_leg(wl, api, exact, steps, world, dev, value, args):
    """The same steps again on ONE stream so that kernels do not overlap, with per-stage HIP events recorded on the
    launch stream (each blend / per-Gaussian stage is exactly one kernel launch)."""
    from luciddreamer_amd import _lib
    step = wl.make_step(api, exact, 1, not args.no_fused_accumulate)
    step()
    _lib.profile_enable(True)
    dt_prof, _ = timed(step, steps, world, dev)
    stages = _lib.profile_read()
    _lib.profile_enable(False)
    wl.finish()
    P, V_mean, R_mean, N, T, K, M, V = wl.P, wl.V_mean, wl.R_mean, wl.N, wl.T, wl.K, wl.M, wl.V
    single_kernel = ("preprocess", "render_fwd", "render_bwd", "gauss_bwd")
    dom = max(single_kernel, key=lambda s: stages[s][0])
    dom_ms, dom_calls = stages[dom]
    dom_avg_s = dom_ms / max(dom_calls, 1) * 1e-3
    dom_bytes = stage_bytes(dom, P, V_mean, R_mean, N, T, K, M)
    achieved = dom_bytes / dom_avg_s / 1e9
    b_f, b_b = path_bytes(P, V_mean, R_mean, N, T, K, M)
    b_moved = moved_bytes(P, V_mean, R_mean, N, T, K, M, V)
    per_rank_views_s = value / world
    # HBM traffic of the dominant kernel: NOT measured in this run (rocprofv3 --pmc needs its own passes).  It is read
    # from the committed counter file of the same workload and build generation (tools/pmc_run.sh: separate passes for
    # FETCH_SIZE and WRITE_SIZE, both in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane
    # reads on gfx950); `traffic_source` says which file.
    traffic, valu, source = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_c3.json")
    if wl.name == "c3" and args.gaussians is None and os.path.exists(pmc_path):
        try:
            allpmc = json.load(open(pmc_path))
            pmc = allpmc.get("k_" + dom) or allpmc.get("k_" + dom + "<false>") or {}
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
                source = "committed PMC pass profiles/pmc_c3.json" + (f" ({allpmc['_collected']})" if "_collected" in allpmc else "")
            if "SQ_INSTS_VALU" in pmc:
                # the blend kernels are VALU bound: wave-instructions per launch from the PMC
                # pass over this launch's measured duration.  Reference rates (tools/valu_microbench.hip on MI355X):
                # 912 G wave-instr/s for v_fma_f32, 453 G/s for v_pk_fma_f32
                ginst = pmc["SQ_INSTS_VALU"] / dom_avg_s / 1e9
                valu = {"wave_insts_per_launch": int(pmc["SQ_INSTS_VALU"]), "achieved_ginst_s": round(ginst, 1),
                        "reference_ginst_s": {"v_fma_f32": 912.0, "v_pk_fma_f32": 453.0}}
                if "SQ_ACTIVE_INST_VALU" in pmc:
                    # cycles in which a SIMD's VALU was executing (counter is in units of 4 cycles, summed over the
                    # 1024 SIMDs) over the cycles of this launch at the 2.4 GHz peak clock
                    valu["valu_busy_frac"] = round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * dom_avg_s * 2.4e9), 4)
        except (OSError, ValueError):
            traffic = None
    return {
        "bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source, "valu_issue": valu,
        "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 4),
        "launches": dom_calls,
        # whole path per view against the HBM peak: with the section-8d byte model as written (`contract`: it prices the
        # reference's 300 B/Gaussian gradient zero-fill, which this design does not perform) and with the bytes this
        # design has to move (`moved`)
        "path_bytes_per_view": int(b_f + b_b),
        "path_frac_contract": round((b_f + b_b) * per_rank_views_s / (HBM_PEAK_GBS * 1e9), 5),
        "path_bytes_moved_per_view": int(b_moved),
        "path_frac_moved": round(b_moved * per_rank_views_s / (HBM_PEAK_GBS * 1e9), 5),
        "stage_ms_per_view": {k: round(v[0] / max(V * steps, 1), 4) for k, v in stages.items()},
        "instrumented_views_per_s": round(world * V * steps / dt_prof, 2),
        "measured_with_streams": 1,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--views", type=int, default=None, help="views per rank per step")
    ap.add_argument("--gaussians", type=int, default=None, help="override the Gaussian count (debug)")
    ap.add_argument("--exact", action="store_true", help="reference-style host sync per view instead of async mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the entry_points / other_workloads legs")
    ap.add_argument("--resolution", default=None, help="WxH override (the metric uses 1920x1080)")
    ap.add_argument("--sh-degree", type=int, default=3, choices=[0, 1, 2, 3], help="active SH degree (diagnostics; the metric uses 3)")
    ap.add_argument("--api", default="views", choices=["views", "autograd", "views-loss"],
                    help="views: one lr_views_accumulate call per step (parallel.ViewBatch); autograd: the drop-in "
                         "GaussianRasterizer autograd op per view (parallel.ViewStreams); views-loss: the views step with "
                         "the fused L1+DSSIM loss against a target image formed inside (not the metric: extra work)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the views of a step alternate on (forward of view i+1 overlaps backward of view i)")
    ap.add_argument("--no-fused-accumulate", action="store_true",
                    help="let autograd accumulate dense per-view gradients instead of the in-kernel accumulation")
    args = ap.parse_args()

    from luciddreamer_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    rank, world, dev = parallel.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    backend = torch.distributed.get_backend() if world > 1 else None
    backend_world = torch.distributed.get_world_size() if world > 1 else 1

    res = tuple(int(v) for v in args.resolution.lower().split("x")) if args.resolution else None
    wl = Workload(args.workload, args, rank, world, dev, args.gaussians, args.views, res)
    cfg = {"workload": wl.label, "views_per_rank_per_step": wl.V, "gaussians": wl.P, "visible_mean": round(wl.V_mean, 1),
           "num_rendered_mean": round(wl.R_mean, 1), "sh_degree": args.sh_degree, "resolution": [wl.W, wl.H],
           "grad_bucket_bytes": int(wl.grads.flat.numel() * 4)}
    fused = not args.no_fused_accumulate
    value, ms_per_step, host_issue_ms, _ = run_leg(wl, args.api, args.exact, args.streams, args.steps, args.warmup, world, dev, fused)
    roofline = roofline_leg(wl, args.api, args.exact, args.steps, world, dev, value, args)

    extras = rank == 0 and world == 1 and not args.no_extras and args.workload == "c3" and args.api == "views" \
        and not args.exact and args.gaussians is None and args.views is None and res is None
    entry_points, other = None, None
    if extras:
        entry_points = {}
        for key, api, exact in (("drop_in_views_per_s", "autograd", False), ("exact_mode_views_per_s", "autograd", True),
                                ("views_loss_views_per_s", "views-loss", False)):
            v, ms, host, _ = run_leg(wl, api, exact, args.streams, args.steps, args.warmup, world, dev, fused)
            entry_points[key] = round(v, 1)
            entry_points[key.replace("_views_per_s", "_host_issue_ms_per_step")] = round(host, 3)
        entry_points["note"] = ("drop_in: GaussianRasterizer autograd op per view (the reference's API), async mode, "
                                f"{args.streams} streams; exact_mode: the same with the reference's host round trip per view; "
                                "views_loss: the headline step with the fused L1+DSSIM loss formed inside")
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(wl.cloud, wl.my_cams, wl.degree, wl.H, wl.W)
    if extras:
        other = {}
        del wl
        torch.cuda.empty_cache()
        for name in ("c3box", "c4shape"):
            w2 = Workload(name, args, rank, world, dev)
            steps2 = max(2, min(args.steps, 5))
            v, ms, host, _ = run_leg(w2, "views", False, args.streams, steps2, 1, world, dev, fused)
            b_f, b_b = path_bytes(w2.P, w2.V_mean, w2.R_mean, w2.N, w2.T, w2.K, w2.M)
            b_m = moved_bytes(w2.P, w2.V_mean, w2.R_mean, w2.N, w2.T, w2.K, w2.M, w2.V)
            other[name] = {"workload": w2.label, "value": round(v, 1), "unit": "views/s", "steps": steps2,
                           "ms_per_step": round(ms, 3), "visible_mean": round(w2.V_mean, 1),
                           "num_rendered_mean": round(w2.R_mean, 1),
                           "path_frac_contract": round((b_f + b_b) * v / (HBM_PEAK_GBS * 1e9), 5),
                           "path_frac_moved": round(b_m * v / (HBM_PEAK_GBS * 1e9), 5)}
            del w2
            torch.cuda.empty_cache()

    if rank == 0:
        metric = {"c2": "views/sec fwd+bwd @1080p (1e5 Gaussians)", "c4shape": "views/sec fwd+bwd @1440p (3e6 Gaussians)",
                  "c5shape": "views/sec fwd+bwd @512x512 (1e6 Gaussians)"}.get(args.workload, "views/sec fwd+bwd @1080p (1e6 Gaussians)")
        cfg.update({"mode": "exact (host sync per view)" if args.exact else "async (no host sync per view)",
                    "parallelism": f"dp{world} (views sharded; gradients all-reduced in {parallel.REDUCE_CHUNKS} chunks, the first "
                                   "under the second half of the views)" if world > 1 else "dp1",
                    "dist_backend": backend, "dist_world_size": backend_world, "streams_per_rank": args.streams,
                    "api": args.api, "host_issue_ms_per_step": round(host_issue_ms, 3)})
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": cfg,
            "roofline": roofline,
            "entry_points": entry_points,
            "other_workloads": other,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def run_cpu_baseline(cloud, cam, degree, H, W):
    """Time the CPU oracle (the 'port' of the reference semantics, pinned bit for bit to the reference's own sources by
    tests/test_oracle_ref.py; the reference has no CPU path) on up to 10 views fwd+bwd of the same workload (bounded to
    ~20 s), all host cores (OpenMP); reports the median."""
    import numpy as np
    from luciddreamer_amd import synthetic
    from oracle import oracle
    n = lambda t: t.detach().cpu().numpy()
    g = n(synthetic.upstream_grad(H, W))
    oracle.lib()
    cams = cam if isinstance(cam, (list, tuple)) else [cam]

    def one_view(c):
        tfx, tfy = math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5)
        t0 = time.perf_counter()
        res = oracle.forward(np.zeros(3, np.float32), n(cloud["means3D"]), None, n(cloud["opacities"]), n(cloud["scales"]),
                             n(cloud["rotations"]), 1.0, None, n(c.world_view_transform), n(c.full_proj_transform),
                             tfx, tfy, H, W, n(cloud["shs"]), degree, n(c.camera_center))
        t1 = time.perf_counter()
        oracle.backward(res, g)
        return t1 - t0, time.perf_counter() - t1

    one_view(cams[0])                                   # warm-up (thread pool, page faults)
    times = []
    budget_t0 = time.perf_counter()
    for i in range(10):                                 # up to 10 views of the path, bounded to ~20 s of CPU time
        times.append(one_view(cams[i % len(cams)]))
        if time.perf_counter() - budget_t0 > 20.0:
            break
    tot = sorted(f + b for f, b in times)
    med = tot[len(tot) // 2]
    fwd_med = sorted(f for f, _ in times)[len(times) // 2]
    cores = os.cpu_count() or 1
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": round(1.0 / med, 4), "unit": "views/s", "cores": cores, "kind": "port",
           "sample": f"median of {len(times)} views fwd+bwd of the same workload after 1 warm-up view "
                     f"({fwd_med:.2f}s fwd + {med - fwd_med:.2f}s bwd), OpenMP over {cores} host threads",
           "cpu_model": model}
    # Second baseline of the same leg, when the build container shipped it: the reference's OWN kernels (forward.cu,
    # backward.cu, rasterizer_impl.cu compiled by hipcc for gfx950, oracle/build_ref.py build_device()) on this GPU, driven
    # the way its binding drives them: zero-filled outputs and gradients, one blocking read-back per forward, legacy stream.
    try:
        from oracle import ref_device
        if ref_device.available():
            dev = torch.device("cuda:0")
            c = {k: v.to(dev).contiguous() for k, v in cloud.items()}
            gd = torch.from_numpy(g).to(dev)
            bg = torch.zeros(3, device=dev)
            r = ref_device.Renderer()
            cd = [x.to(dev) for x in cams]

            def ref_view(x):
                r.forward(bg, c["means3D"], None, c["opacities"], c["scales"], c["rotations"], 1.0, None,
                          x.world_view_transform.contiguous(), x.full_proj_transform.contiguous(), math.tan(x.FoVx * 0.5),
                          math.tan(x.FoVy * 0.5), H, W, c["shs"], degree, x.camera_center.contiguous(), sync=False)
                r.backward(gd, sync=False)
            for x in cd[:3]:
                ref_view(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _ in range(3):
                for x in cd:
                    ref_view(x)
                    n += 1
            ref_device.lib().refdev_sync()
            dt = time.perf_counter() - t0
            out["reference_kernels_on_this_gpu"] = {
                "value": round(n / dt, 1), "unit": "views/s",
                "what": "the reference's own CUDA sources compiled by hipcc for gfx950 (oracle/_ref, -O3 -ffp-contract=off, hipCUB sort/scan), "
                        f"{n} views fwd+bwd of the same workload, one stream, its own host read-back per forward"}
    except Exception as e:                                   # the baseline is optional evidence, never a reason to fail
        out["reference_kernels_on_this_gpu"] = {"error": str(e)[:200]}
    return out


if __name__ == "__main__":
    main()
