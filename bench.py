#!/usr/bin/env python
"""bench.py -- views/s forward+backward of the MI355X Gaussian-splat rasterizer (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2] [--views V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic views: every rank renders its V
views of the camera path forward+backward (rasterizer autograd op, upstream gradient dL/dcolor =
N(0,1)), gradients accumulate in one flat fp32 bucket, and (N > 1) the bucket is all-reduced once
over RCCL.  Weak scaling: V views per rank per step, N*V distinct views per step.
Inputs are resident in HBM before the timed region; data is synthetic.

Workloads (BASELINE.json configs):
  c3 (default; the metric's configuration): 1e6 Gaussians, SH degree 3, 1920x1080, band cloud,
      rotate360 camera path, V = 30 views per rank per step.
  c2: 1e5 Gaussians, SH degree {degree}, {W}x{H}, box cloud, single identity view repeated V times.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library
on the launch stream) and `cpu_baseline` (the CPU oracle = "port" of the reference semantics, timed
on this box's host cores on one view of the same workload).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c3box"])
    ap.add_argument("--views", type=int, default=None, help="views per rank per step")
    ap.add_argument("--gaussians", type=int, default=None, help="override the Gaussian count (debug)")
    ap.add_argument("--exact", action="store_true", help="reference-style host sync per view instead of async mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resolution", default="1920x1080", help="WxH (the metric uses 1920x1080)")
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

    from luciddreamer_amd import _C, _lib, cameras, config, parallel, synthetic
    from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    rank, world, dev = parallel.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    W, H = (int(v) for v in args.resolution.lower().split("x"))
    degree = args.sh_degree
    if args.workload == "c3":
        P = args.gaussians or 1_000_000
        V = args.views or 30
        cloud = synthetic.make_cloud(P, "band", 0)
        path = cameras.rotate360_path(W, H, n_views=V * world)
        my_cams = [path[i] for i in parallel.shard_views(len(path), rank, world)]
        wl_name = f"C3: {P} Gaussians, SH degree {degree}, {W}x{H}, band cloud, rotate360 path, {V} views/rank/step"
    else:
        P = args.gaussians or (100_000 if args.workload == "c2" else 1_000_000)
        V = args.views or 30
        cloud = synthetic.make_cloud(P, "box", 0)
        my_cams = [cameras.identity_camera(W, H)] * V
        tag = "C2" if args.workload == "c2" else "C3-box (all Gaussians in front of the camera)"
        wl_name = f"{tag}: {P} Gaussians, SH degree {degree}, {W}x{H}, box cloud, identity view x{V}/rank/step"
    M = cloud["shs"].shape[1]
    K = NCOEF[degree]
    N = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)

    leaf = {k: v.to(dev).requires_grad_(True) for k, v in cloud.items()}
    params = [leaf["means3D"], leaf["scales"], leaf["rotations"], leaf["opacities"], leaf["shs"]]
    grads = parallel.FlatGrads(params)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    grad_color = synthetic.upstream_grad(H, W).to(dev)
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in my_cams]
    rasterizers = []
    for c in cams:
        rs = GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, 1.0,
                                           c.world_view_transform, c.full_proj_transform, degree, c.camera_center,
                                           False, False)
        rasterizers.append(GaussianRasterizer(rs))

    # per-view R (= num_rendered) and V (= visible count), measured once in exact mode; also seeds the
    # async-mode binning capacity so no view in the timed region needs a host round trip
    empty = torch.Tensor([])
    view_stats = []
    seen = {}
    with torch.no_grad():
        for c, r in zip(cams, rasterizers):
            key = id(c.world_view_transform)
            if key not in seen:
                rs = r.raster_settings
                out = _C.rasterize_gaussians(bg, leaf["means3D"], empty, leaf["opacities"], leaf["scales"],
                                             leaf["rotations"], 1.0, empty, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                             rs.tanfovy, H, W, leaf["shs"], degree, rs.campos, False, False)
                seen[key] = (int(out[0]), int((out[3] > 0).sum().item()))
            view_stats.append(seen[key])
    R_mean = sum(s[0] for s in view_stats) / len(view_stats)
    V_mean = sum(s[1] for s in view_stats) / len(view_stats)
    config.reset()
    config.set_fused_grad_accumulation(not args.no_fused_accumulate)
    if not args.exact:
        config.set_async(True, headroom=1.25)
        config._hwm[(dev.index, P, H, W)] = max(s[0] for s in view_stats)

    m2d_grad = torch.zeros(P, 3, device=dev)

    pipe = parallel.ViewStreams(dev, args.streams)
    batch = None
    if args.api in ("views", "views-loss"):
        cap = int(max(s[0] for s in view_stats) * 1.25) + 4096
        if args.api == "views":
            batch = parallel.ViewBatch(cams, [grad_color] * len(cams), degree, bg, cap, n_streams=args.streams)
        else:
            target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
            batch = parallel.ViewBatch(cams, None, degree, bg, cap, n_streams=args.streams, targets=[target] * len(cams))
        acc = {"means3D": leaf["means3D"].grad, "means2D": m2d_grad, "opacity": leaf["opacities"].grad,
               "sh": leaf["shs"].grad, "scales": leaf["scales"].grad, "rotations": leaf["rotations"].grad}

    def step():
        grads.zero_()
        if batch is not None:                       # one C call: all views, fwd+bwd, accumulate in place
            m2d_grad.zero_()
            batch.run(leaf["means3D"], leaf["opacities"], leaf["scales"], leaf["rotations"], leaf["shs"], acc)
            grads.all_reduce()
            return
        means2D.grad = m2d_grad.zero_() if not args.no_fused_accumulate else None
        pipe.begin_step()
        for r in rasterizers:
            pipe.run_view(
                lambda r=r: r(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"],
                              shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])[0],
                lambda color: color.backward(grad_color))
        pipe.end_step()
        grads.all_reduce()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    host_issue = [0.0]

    def timed(n_steps):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        host_issue[0] = (time.perf_counter() - t0) / n_steps * 1e3      # host time to ISSUE a step (no device sync yet)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    dt = timed(args.steps)
    host_issue_ms = host_issue[0]
    config.drain()
    if batch is not None:
        batch.check()
    views_total = world * V * args.steps
    value = views_total / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline leg: the same steps again, on ONE stream so that kernels do not overlap, with per-stage
    # HIP events recorded on the launch stream (each blend / per-Gaussian stage is exactly one kernel launch)
    if batch is not None:
        batch.n_streams = 1
    pipe = parallel.ViewStreams(dev, 1)
    step()
    _lib.profile_enable(True)
    dt_prof = timed(args.steps)
    stages = _lib.profile_read()
    _lib.profile_enable(False)
    config.drain()
    single_kernel = ("preprocess", "render_fwd", "render_bwd", "gauss_bwd")
    dom = max(single_kernel, key=lambda s: stages[s][0])
    dom_ms, dom_calls = stages[dom]
    dom_avg_s = dom_ms / max(dom_calls, 1) * 1e-3
    dom_bytes = stage_bytes(dom, P, V_mean, R_mean, N, T, K, M)
    achieved = dom_bytes / dom_avg_s / 1e9
    b_f, b_b = path_bytes(P, V_mean, R_mean, N, T, K, M)
    per_rank_views_s = value / world
    # HBM traffic of the dominant kernel from the committed PMC passes (tools/pmc_run.sh: separate rocprofv3 --pmc
    # runs for FETCH_SIZE and WRITE_SIZE; both in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    # 16-byte-per-lane reads on gfx950).  Only meaningful for the workload the counters were collected on.
    traffic = None
    valu = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_c3.json")
    if args.workload == "c3" and args.gaussians is None and os.path.exists(pmc_path):
        try:
            allpmc = json.load(open(pmc_path))
            pmc = allpmc.get("k_" + dom) or allpmc.get("k_" + dom + "<false>") or {}
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
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
    roofline = {
        "bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_issue": valu,
        "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 4),
        "launches": dom_calls,
        "path_bytes_per_view": int(b_f + b_b),
        "path_frac_of_hbm_peak": round((b_f + b_b) * per_rank_views_s / (HBM_PEAK_GBS * 1e9), 5),
        "stage_ms_per_view": {k: round(v[0] / max(V * args.steps, 1), 4) for k, v in stages.items()},
        "instrumented_views_per_s": round(world * V * args.steps / dt_prof, 2),
        "measured_with_streams": 1,
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(cloud, my_cams, degree, H, W)

    if rank == 0:
        line = {
            "metric": "views/sec fwd+bwd @1080p (1e6 Gaussians)" if args.workload != "c2" else "views/sec fwd+bwd @1080p (1e5 Gaussians)",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_name, "views_per_rank_per_step": V, "gaussians": P, "visible_mean": round(V_mean, 1),
                       "num_rendered_mean": round(R_mean, 1), "sh_degree": degree, "resolution": [W, H],
                       "mode": "exact (host sync per view)" if args.exact else "async (no host sync per view)",
                       "parallelism": f"dp{world} (views sharded, one flat grad all-reduce/step)", "streams_per_rank": args.streams, "api": args.api,
                       "grad_bucket_bytes": int(grads.flat.numel() * 4),
                       "host_issue_ms_per_step": round(host_issue_ms, 3)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def run_cpu_baseline(cloud, cam, degree, H, W):
    """Time the CPU oracle (the 'port' of the reference semantics; the reference has no CPU path and its
    CUDA sources cannot be built here) on up to 10 views fwd+bwd of the same workload (bounded to ~20 s), all host
    cores (OpenMP); reports the median."""
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
    return {"value": round(1.0 / med, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"median of {len(times)} views fwd+bwd of the same workload after 1 warm-up view "
                      f"({fwd_med:.2f}s fwd + {med - fwd_med:.2f}s bwd), OpenMP over {cores} host threads",
            "cpu_model": model}


if __name__ == "__main__":
    main()
