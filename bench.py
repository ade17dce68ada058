#!/usr/bin/env python
"""bench.py -- views/s forward+backward of the MI355X Gaussian-splat rasterizer (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c3box|c4shape|c5shape] [--views V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic views: every rank renders its views of the camera
path forward+backward (upstream gradient dL/dcolor = N(0,1)), gradients accumulate in one flat fp32 bucket, and (N > 1)
the bucket is all-reduced over RCCL, joined before the step ends.  Inputs are resident in HBM before the timed region;
data is synthetic.

Scaling.  The metric's configuration (BASELINE.json config 3) is "rotate360 path (30 views) data-parallel over the GPUs
of one node with one gradient all-reduce": the STEP is 30 views whatever N is.  `--scaling strong` (the default for N > 1)
runs exactly that: view i of the 30 goes to rank i mod N, every rank accumulates its 30/N views and the 236 B/Gaussian
bucket is all-reduced once per step -- total work fixed, "scaling": "strong".  `--scaling weak` keeps 30 views PER RANK
(N x 30 distinct views per step), the round-1/2 behaviour.

Workloads (BASELINE.json configs):
  c3 (default; the metric's configuration): 1e6 Gaussians, SH degree 3, 1920x1080, band cloud, rotate360 camera path,
      V = 30 views per rank per step.
  c2: 1e5 Gaussians, box cloud, identity view.  c3box: the dense reading of C3 (all 1e6 Gaussians in view).
  c4shape: 3e6 Gaussians at 2560x1440, all in view.  c5shape: 1e6 Gaussians at 512x512, all in view.
  ld512: 1e6 pixel-sized Gaussians in one layer on a panorama band, 512x512 views of the rotate360 path (LucidDreamer's own
      scene statistics: a quarter of the cloud in view, about one Gaussian per pixel).

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
import traceback

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
    # LucidDreamer's own scene statistics (luciddreamer_amd/synthetic.py kind "shell"): one layer of pixel-sized Gaussians
    # lifted from a panorama, a quarter of them in a 512 x 512 view of the rotate360 path, about one per pixel
    "ld512": ("shell", 1_000_000, (512, 512), 30, "LD-512 (one layer of pixel-sized Gaussians on a panorama band)"),
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


def stage_bytes_moved(stage, P, V, R, N, T, K, M):
    """stage_bytes without what this design never moves: the per-Gaussian backward's model prices the reference's gradient
    zero-fill P (108 + 12 M) (RAST/rasterize_points.cu:154-162); here visible rows are read-modify-written instead (236 V)."""
    b = stage_bytes(stage, P, V, R, N, T, K, M)
    return b - P * (108 + 12 * M) + 236 * V if stage == "gauss_bwd" else b


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

    def __init__(self, name, args, rank, world, dev, gaussians=None, views=None, resolution=None, scaling="weak"):
        from luciddreamer_amd import _C, cameras, parallel, synthetic
        from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        kind, P, (W, H), V, tag = WORKLOADS[name]
        P, V = gaussians or P, views or V
        if resolution:
            W, H = resolution
        self.name, self.P, self.W, self.H, self.V, self.dev, self.world = name, P, W, H, V, dev, world
        self.args = args
        self.degree = args.sh_degree
        cloud = synthetic.make_cloud(P, kind, 0)
        # views of one step over all ranks: V per rank (weak) or V in total, view i -> rank i mod world (strong)
        total = V * world if scaling == "weak" else V
        mine = parallel.shard_views(total, rank, world)
        per = f"{V} views/rank/step" if scaling == "weak" else f"{V} views/step over {world} rank(s)"
        if kind in ("band", "shell"):
            path = cameras.rotate360_path(W, H, n_views=total)
            my_cams = [path[i] for i in mine]
            self.label = f"{tag}: {P} Gaussians, SH degree {self.degree}, {W}x{H}, {kind} cloud, rotate360 path, {per}"
        else:
            my_cams = [cameras.identity_camera(W, H)] * len(mine)
            self.label = f"{tag}: {P} Gaussians, SH degree {self.degree}, {W}x{H}, box cloud, identity view, {per}"
        if not my_cams:
            raise SystemExit(f"rank {rank} has no view: {total} views per step over {world} ranks")
        self.V, self.total_views, self.scaling = len(my_cams), total, scaling
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
        self.batch, self.chunks = None, 1

    def make_step(self, api, exact, streams, fused=True):
        """Returns step(): one optimisation step's worth of views through the given entry point."""
        from luciddreamer_amd import config, parallel
        leaf, dev = self.leaf, self.dev
        config.reset()
        config.set_fused_grad_accumulation(fused)
        # the drop-in operator runs with the library's DEFAULTS (async after two exact warm calls per problem size; the
        # views of a ViewStreams pipeline use the non-waiting overflow policy by themselves); `exact` = the reference's
        # host round trip in every forward
        config.set_async(not exact)
        self.batch = None
        if api in ("views", "views-loss"):
            named = {"means3D": leaf["means3D"], "scales": leaf["scales"], "rotations": leaf["rotations"],
                     "opacity": leaf["opacities"], "sh": leaf["shs"]}
            # ONE all-reduce after the rank's last view.  (Rounds 2-4 split a rank's views into two groups and ran the first
            # group's all-reduce under the second group: every group reduces a FULL bucket -- each view touches rows of every
            # tensor -- so the step is c1 + max(ar, c2) + ar against c + ar unsplit: never shorter, and twice the bytes on the
            # links when the step is communication bound.  parallel.ChunkedViewStep still takes chunks = 2 for A/B runs.)
            self.chunks = chunks = 1
            if api == "views" and getattr(self.args, "exchange", "allreduce") == "sharded-adam":
                # the step as a TRAINING step (not the metric: it adds the optimizer): no all-reduce -- reduce-scatter of the
                # bucket, Adam on this rank's shard, all-gather of the parameters (parallel.ShardedAdam)
                ordered = [torch.nn.Parameter(named[k].detach()) for k in parallel.ChunkedViewStep.ORDER]
                named = dict(zip(parallel.ChunkedViewStep.ORDER, ordered))
                bucket = parallel.ShardedAdam.make_buckets(ordered)
                self.batch = parallel.ChunkedViewStep(self.cams, [self.grad_color] * len(self.cams), named, self.degree,
                                                      self.bg, self.capacity, n_streams=streams, chunks=1, grads=bucket)
                self.opt = parallel.ShardedAdam(ordered, bucket, [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3])
                self.grads = self.batch.grads
                batch, opt = self.batch, self.opt

                def step():
                    self.m2d_grad.zero_()
                    batch.run(self.m2d_grad, reduce=False)
                    opt.step()
                return step
            if api == "views" and getattr(self.args, "exchange", "allreduce") == "split-sharded-adam":
                # the same training step with the parameter all-gather split (parallel.SplitShardedAdam): geometry (44 B per
                # Gaussian) joined inside the optimizer step, the SH gather (192 B) left in flight until the next step's first
                # forward needs it -- here: until the bucket is zeroed and the views are issued again
                ordered = {k: torch.nn.Parameter(named[k].detach()) for k in parallel.ChunkedViewStep.ORDER}
                lrs = dict(zip(parallel.ChunkedViewStep.ORDER, [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3]))
                self.opt = parallel.SplitShardedAdam(ordered, lrs)
                self.batch = parallel.ChunkedViewStep(self.cams, [self.grad_color] * len(self.cams), ordered, self.degree,
                                                      self.bg, self.capacity, n_streams=streams, chunks=1, grads=self.opt.grads)
                self.exchange_phases = self.opt.bytes_per_step
                batch, opt = self.batch, self.opt

                def step():
                    self.m2d_grad.zero_()          # (everything that reads geometry only could be issued here, before the join)
                    opt.wait()                     # the SH rows of the last step's gather: the forward reads them
                    batch.run(self.m2d_grad, reduce=False)
                    opt.step()
                return step
            if api == "views" and getattr(self.args, "exchange", "allreduce") == "sparse-rows":
                # the same sum as the all-reduce; only the rows this rank's views touched travel in the reduce half
                self.chunks = 1
                self.batch = parallel.ChunkedViewStep(self.cams, [self.grad_color] * len(self.cams), named, self.degree,
                                                      self.bg, self.capacity, n_streams=streams, chunks=1)
                self.grads = self.batch.grads
                batch = self.batch
                self.exchange_info = []

                def step():
                    self.m2d_grad.zero_()
                    batch.run(self.m2d_grad, reduce=False)
                    self.exchange_info.append(parallel.sparse_rows_all_reduce(batch.grads.views))
                return step
            if api == "views":
                self.batch = parallel.ChunkedViewStep(self.cams, [self.grad_color] * len(self.cams), named, self.degree,
                                                      self.bg, self.capacity, n_streams=streams, chunks=chunks)
            else:
                target = torch.rand(3, self.H, self.W, generator=torch.Generator().manual_seed(2)).to(dev)
                self.batch = parallel.ChunkedViewStep(self.cams, None, named, self.degree, self.bg, self.capacity,
                                                      n_streams=streams, targets=[target] * len(self.cams), chunks=chunks)
            self.grads = self.batch.grads                   # the parameters' .grad now live in this step's bucket
            batch = self.batch

            def step():
                self.m2d_grad.zero_()
                batch.run(self.m2d_grad)
            return step
        grads = self.grads = parallel.FlatGrads(self.params)
        pipe = self.pipe = parallel.ViewStreams(dev, streams)
        means2D, grad_color = self.means2D, self.grad_color

        def step():
            grads.zero_()
            means2D.grad = self.m2d_grad.zero_() if fused else None
            pipe.begin_step()
            for r in self.rasterizers:
                # grad_output: the views of a group share one pass of the autograd engine (parallel.ViewStreams.run_view)
                pipe.run_view(
                    lambda r=r: r(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"],
                                  shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])[0],
                    grad_output=grad_color)
            pipe.end_step()
            grads.all_reduce()
        return step

    def finish(self):
        from luciddreamer_amd import config
        config.drain()
        if self.batch is not None:
            self.batch.check()


def timed(step, n_steps, world, dev, collect=True):
    """K steps between barrier + synchronize on both sides; MAX over ranks.  Returns (seconds, host ms to issue a step)."""
    # before the clock: what the set-up of this leg created is still in the Python collector's young generations, and the first
    # collection inside a timed region of a few milliseconds would walk all of it (50-90 ms: profiles/r04q_loop_drift.txt).
    # run_leg collects BEFORE its warm-up steps instead (collect=False here): a collection between the warm-up and the clock
    # leaves the GPU idle for those 50-90 ms, and the first steps after an idle stretch run below the sustained clock
    # (profiles/r06f_step_series.txt: +20 / +9 / +6 / +5 % on the first four steps after 2 s of idle)
    import gc
    if collect:
        gc.collect()
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
    import gc
    step = wl.make_step(api, exact, streams, fused)
    gc.collect()                                    # (see timed())
    for _ in range(warmup):
        step()
    dt, host_ms = timed(step, steps, world, dev, collect=False)
    wl.finish()
    return wl.total_views * steps / dt, dt / steps * 1e3, host_ms, step


def read_counter_csv(directory, counter, acc):
    """Add the rows of `counter` in rocprofv3's *counter_collection.csv files under `directory` to acc[kernel][counter] =
    [sum, launches] (kernel = the k_name<...> part of the mangled name, as tools/pmc_summary.py shortens it).  Returns the
    number of rows read."""
    import csv
    import glob
    import re
    rows = 0
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", row["Kernel_Name"])
                a = acc.setdefault(m.group(1) if m else row["Kernel_Name"][:40], {}).setdefault(counter, [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                rows += 1
    return rows


def pick_kernel(counters, stage, prefer=None):
    """The counter record of the kernel a stage timing belongs to.  `prefer`: a name prefix to take first (the blend backward
    has several shapes: "k_render_bwd_tile" is what the multi-stream headline launches at 1080p, "k_render_bwd<" what a lone
    view gets); otherwise k_<stage><...> before k_<stage>_other ('<' sorts before '_').  Returns (name, record)."""
    names = [k for k in sorted(counters) if isinstance(counters[k], dict) and k.startswith("k_" + stage)]
    if prefer:
        names = [k for k in names if k.startswith(prefer)] + [k for k in names if not k.startswith(prefer)]
    return (names[0], counters[names[0]]) if names else (None, {})


def live_pmc_counters(timeout_s=90):
    """FETCH_SIZE / WRITE_SIZE per kernel and launch, collected NOW on this box: two rocprofv3 passes (one counter each,
    --kernel-trace only, from /tmp with TMPDIR=/tmp -- the recipe of MI355X_MICROARCH.md / tools/pmc_run.sh) over a child run of
    this script on the same workload (4 views, 1 step, no extras).  Returns ({kernel: {counter: mean per launch}}, note); the
    dict is empty when anything goes wrong (no rocprofv3, a pass that fails or does not end in time) and the caller falls back
    to the committed counter file."""
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {}, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {}, "this run is itself being profiled"
    out = tempfile.mkdtemp(prefix="lr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--views", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--no-extras", "--sustain-seconds", "0"]
    acc = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", ctr, "--"] + child
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)              # the process group this call started, nothing else
                p.wait()
                return {}, f"the {ctr} pass did not end within {timeout_s} s"
            if rc != 0:
                return {}, f"the {ctr} pass ended with code {rc}"
            if read_counter_csv(d, ctr, acc) == 0:
                return {}, f"the {ctr} pass wrote no counter rows"
    except (OSError, ValueError, KeyError) as e:
        return {}, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    return ({k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()},
            "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass each, run by this bench.py on this box "
            "(child: bench.py --views 4 --steps 1 --warmup 1 --no-extras)")


def profiled_stages(wl, api, exact, steps, world, dev, args, views_in_flight):
    """K steps on ONE stream (kernels do not overlap) with per-stage HIP events recorded on the launch stream inside the
    library; `views_in_flight` is passed to the library as the hint the multi-stream legs give it (it picks the kernel SHAPE
    of the blend backward: render_bwd.hip blend_shape), so that the kernel timed here is the one the headline launched."""
    from luciddreamer_amd import _lib
    step = wl.make_step(api, exact, 1, not args.no_fused_accumulate)
    _lib.tune_set("views_in_flight", views_in_flight if views_in_flight >= 2 else -1)
    try:
        step()
        _lib.profile_enable(True)
        dt_prof, _ = timed(step, steps, world, dev)
        stages = _lib.profile_read()
        _lib.profile_enable(False)
        wl.finish()
        wl.last_profiled_shapes = _lib.last_launch_shapes()      # (forward, backward) kernel shapes of this pass
    finally:
        _lib.tune_set("views_in_flight", -1)
    return stages, dt_prof


def roofline_leg(wl, api, exact, steps, world, dev, value, args):
    """The same steps again on ONE stream so that kernels do not overlap, with per-stage HIP events recorded on the
    launch stream (each blend / per-Gaussian stage is exactly one kernel launch).  Twice when the headline ran with several
    views in flight: once with the kernel shapes those legs launch (the quoted figures), once as a lone view gets them
    (`lone_view_shape`)."""
    from luciddreamer_amd import _lib
    in_flight = args.streams if api in ("views", "views-loss", "autograd") else 1
    stages, dt_prof = profiled_stages(wl, api, exact, steps, world, dev, args, in_flight)
    shapes = getattr(wl, "last_profiled_shapes", (None, None))
    lone, lone_shapes = None, (None, None)
    if in_flight >= 2:
        lone = profiled_stages(wl, api, exact, steps, world, dev, args, 1)[0]
        lone_shapes = getattr(wl, "last_profiled_shapes", (None, None))
    # name of the blend-backward kernel the pass launched (lr_last_launch_shapes: the rule of render_bwd.hip blend_shape)
    bwd_tile = shapes[1] == "tile"
    bwd_name = "k_render_bwd_tile" if bwd_tile else "k_render_bwd<"
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
    # HBM traffic of the dominant kernel (rocprofv3 --pmc needs its own passes, separate from the timed region): at N = 1 on
    # the default workload the two byte-counter passes are run right here, on this box, over a child run of this script
    # (live_pmc_counters: FETCH_SIZE and WRITE_SIZE, one pass each, both in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md
    # prescribes for 16-byte-per-lane reads on gfx950).  The committed counter file of the same build (tools/pmc_run.sh, more
    # counters: VALU instructions, clocks) supplies `valu_issue`, and the bytes as well where no live pass ran (N > 1, other
    # workloads, --no-extras, a run that is itself being profiled); `traffic_source` says which.
    traffic, valu, source, traffic_ratios = None, None, None, None
    traffic_file, source_file, live, live_note, dom_kernel = None, None, None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_c3.json")
    default_c3 = wl.name == "c3" and args.gaussians is None and args.views is None and args.resolution is None
    if default_c3 and world == 1 and not args.no_extras and not args.no_live_pmc and api == "views" and not exact:
        torch.cuda.synchronize()
        live = live_pmc_counters()
        live_note = live[1]
    if wl.name == "c3" and args.gaussians is None and (os.path.exists(pmc_path) or live):
        try:
            allpmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
            # the counter file is stamped with the lr_version() (a hash of the kernel sources) it was collected on
            # (tools/pmc_run.sh / pmc_summary.py); another build's counters are refused, not quoted
            stamp = allpmc.get("_lr_version")
            if stamp != _lib.lib().lr_version().decode():
                source = f"stale: profiles/pmc_c3.json was collected on '{stamp}', this build is '{_lib.lib().lr_version().decode()}'"
                allpmc = {}
            # the kernel of the single-stream stage timing: the 2-wave k_render_bwd<...>, not the TILE shape the multi-stream legs
            # of the same run launch (k_render_bwd_tile) -- '<' sorts before '_'
            prefer = bwd_name if dom == "render_bwd" else None
            dom_kernel, pmc = pick_kernel(allpmc, dom, prefer)
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
                source = "committed PMC pass profiles/pmc_c3.json" + (f" ({allpmc['_collected']})" if "_collected" in allpmc else "")
            # the byte counters again, collected by THIS run on THIS box (two short rocprofv3 passes over a child run); they
            # replace the committed file's bytes -- which stay in the line as `traffic_committed_file` -- wherever they exist
            if live is not None:
                live_note = live[1]
                for k, v in live[0].items():
                    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                        allpmc.setdefault(k, {})
                        allpmc[k] = dict(allpmc[k], FETCH_SIZE=v["FETCH_SIZE"], WRITE_SIZE=v["WRITE_SIZE"])
                lk, lp = pick_kernel(live[0], dom, prefer)
                dom_kernel = lk or dom_kernel
                if "FETCH_SIZE" in lp and "WRITE_SIZE" in lp:
                    traffic_file, traffic = traffic, int((2.0 * lp["FETCH_SIZE"] + lp["WRITE_SIZE"]) * 1024)
                    source_file, source = source, "live: " + live[1]
            if "SQ_INSTS_VALU" in pmc:
                # the blend kernels are VALU bound: wave-instructions per launch from the PMC pass over
                # this launch's measured duration, against the ARCHITECTURAL issue rate -- a wave64 VALU instruction occupies
                # its SIMD-32 for 2 cycles (MI355X_MICROARCH.md): 1024 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instr/s.  Measured
                # rates of single instructions (tools/valu_microbench.hip, 8 waves per SIMD): v_fma_f32 880 G/s (2.8 cycles),
                # compares / selects / DPP adds ~550 G/s (4.4), v_exp / v_rcp / lane swaps ~270 G/s (9)
                ginst = pmc["SQ_INSTS_VALU"] / dom_avg_s / 1e9
                valu = {"wave_insts_per_launch": int(pmc["SQ_INSTS_VALU"]), "achieved_ginst_s": round(ginst, 1),
                        "architectural_peak_ginst_s": 1228.8, "frac_of_architectural_peak": round(ginst / 1228.8, 4),
                        "measured_single_instruction_ginst_s": {"v_fma_f32": 880.0, "v_pk_fma_f32": 440.0, "v_cmp_or_dpp": 550.0,
                                                                "v_exp_rcp_permlane_swap": 270.0}}
                if "SQ_ACTIVE_INST_VALU" in pmc:
                    # cycles in which a SIMD's VALU was executing (counter is in units of 4 cycles, summed over the
                    # 1024 SIMDs) over the cycles of this launch at the 2.4 GHz peak clock
                    valu["valu_busy_frac"] = round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * dom_avg_s * 2.4e9), 4)
                    valu["simd_cycles_per_instruction"] = round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / pmc["SQ_INSTS_VALU"], 3)
                if "effective_clock_ghz" in pmc:
                    # GRBM_GUI_ACTIVE per XCD over the launch's wall time in the counter pass: the clock the kernel ran at
                    valu["effective_clock_ghz"] = round(pmc["effective_clock_ghz"], 3)
            # counter traffic against the algorithmic bytes, per kernel of the path (wasted re-reads show up here first)
            ratios = {}
            for st in single_kernel:
                # the kernel the steady state runs: the pooled preprocess (the thread-per-Gaussian one serves the warm-up view)
                # ... and the blend kernels of the shape the headline launched (one wave per tile at 1080p with views in flight)
                names = sorted((k for k, v in allpmc.items() if isinstance(v, dict) and k.startswith("k_" + st)),
                               key=lambda k: (0 if "_pool" in k else 1, 0 if ("_tile" in k) == bwd_tile else 1, k))
                pk = allpmc[names[0]] if names else None
                if pk and "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
                    tb = (2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024
                    ab = stage_bytes(st, P, V_mean, R_mean, N, T, K, M)
                    if st == "gauss_bwd":
                        # without the reference's gradient zero-fill P (108 + 12 M), which this design never performs; with
                        # the read-modify-write of the accumulated rows (236 B per visible Gaussian read back)
                        ab = ab - P * (108 + 12 * M) + 236 * V_mean
                    ratios[names[0]] = {"counter_bytes": int(tb), "algorithmic_bytes": int(ab), "ratio": round(tb / ab, 3)}
            traffic_ratios = ratios or None
        except (OSError, ValueError):
            traffic = None
    moved_view_s = b_moved / max(sum(stages[k][0] for k in stages) / max(V * steps, 1) * 1e-3, 1e-12)
    lone_shape = None
    if lone is not None:
        ms, calls = lone[dom]
        avg = ms / max(calls, 1) * 1e-3
        lone_shape = {"kernel": "k_" + dom, "kernel_shapes": {"forward": lone_shapes[0], "backward": lone_shapes[1]},
                      "avg_launch_ms": round(avg * 1e3, 4), "achieved": round(dom_bytes / avg / 1e9, 2),
                      "frac": round(dom_bytes / avg / 1e9 / HBM_PEAK_GBS, 5),
                      "what": "the same stage as a lone view gets it (views_in_flight = 1)"}
    # every single-kernel stage of the path against the HBM peak (algorithmic bytes / its own launch time)
    per_stage = {}
    for st in single_kernel:
        ms, calls = stages[st]
        if calls:
            ab = stage_bytes_moved(st, P, V_mean, R_mean, N, T, K, M)
            per_stage[st] = {"us": round(ms / calls * 1e3, 2), "algorithmic_bytes": int(ab),
                             "frac": round(ab / (ms / calls * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
    return {
        "bound": "hbm", "kernel": dom_kernel or (bwd_name.rstrip("<") if dom == "render_bwd" else "k_" + dom),
        "kernel_is_what_the_headline_launched": True, "views_in_flight_hint": in_flight,
        "kernel_shapes": {"forward": shapes[0], "backward": shapes[1]},
        "lone_view_shape": lone_shape, "per_stage": per_stage,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source,
        "traffic_committed_file": traffic_file, "traffic_committed_file_source": source_file,
        "live_pmc": live_note, "valu_issue": valu,
        "counter_vs_algorithmic_bytes": traffic_ratios,
        "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 4),
        "launches": dom_calls,
        # whole path per view against the HBM peak, with the bytes this design has to move
        "path_bytes_moved_per_view": int(b_moved),
        "path_frac_moved": round(b_moved * per_rank_views_s / (HBM_PEAK_GBS * 1e9), 5),
        # footnote: the section-8d byte model as written also prices the reference's 300 B/Gaussian gradient zero-fill, which
        # this design does not perform -- not work done, not a figure to quote
        "footnote_contract_model": {"path_bytes_per_view": int(b_f + b_b),
                                    "path_frac": round((b_f + b_b) * per_rank_views_s / (HBM_PEAK_GBS * 1e9), 5)},
        "stage_ms_per_view": {k: round(v[0] / max(V * steps, 1), 4) for k, v in stages.items()},
        "stage_sum_ms_per_view": round(sum(v[0] for v in stages.values()) / max(V * steps, 1), 4),
        "path_frac_moved_single_stream": round(moved_view_s / (HBM_PEAK_GBS * 1e9), 5),
        "instrumented_views_per_s": round(wl.total_views * steps / dt_prof, 2),
        "measured_with_streams": 1,
    }


def event_ms(fn, reps, warm=2):
    """Average milliseconds per call of fn() by HIP events on the current stream (every kernel these helpers time is
    launched on torch's current stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def measure_f_rows(dev):
    """SURVEY.md section 8f rows in numbers (outside every timed region of the headline): the fused L1 + DSSIM loss
    (/root/reference/utils/loss.py:18-69) at 512^2 and 1080p, distCUDA2 (simple_knn.cu:186-221) at 1 M / 3 M points, the
    one-launch Adam step and one prune of the parameter set (scene/gaussian_model.py:273-304: 6 parameters + 12 Adam
    moments + 3 statistics) at C4 size.  HIP-event averages; bytes are the algorithmic minimum of each op, `frac` against 8 TB/s."""
    out = {}
    try:
        from luciddreamer_amd import _lib
        from luciddreamer_amd.loss import l1_dssim_loss
        L = _lib.lib()
        for W, H in ((512, 512), (1920, 1080)):
            gt = torch.rand(3, H, W, device=dev)
            img = (0.7 * gt + 0.3 * torch.rand(3, H, W, device=dev)).requires_grad_(True)
            # the two kernels through the C-ABI, back to back on the current stream (no autograd, no allocation): device time
            x = img.detach()
            out3, up, grad = torch.empty(3, device=dev), torch.ones(1, device=dev), torch.empty_like(x)
            ws = torch.empty((L.lr_loss_workspace_bytes(3, H, W),), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream(dev).cuda_stream

            def kernels():
                L.lr_l1_dssim_forward(3, H, W, x.data_ptr(), gt.data_ptr(), 0.2, out3.data_ptr(), ws.data_ptr(), ws.numel(), st)
                L.lr_l1_dssim_backward(3, H, W, x.data_ptr(), gt.data_ptr(), 0.2, up.data_ptr(), ws.data_ptr(), grad.data_ptr(), st)
            ms = event_ms(kernels, 200, warm=5)

            def fb():
                img.grad = None
                l1_dssim_loss(img, gt, 0.2).backward()
            ms_py = event_ms(fb, 30)
            # forward reads image + target, backward reads both again and writes the gradient: 5 planes of 3 H W floats
            # (the workspace of window sums the backward re-reads is this design's own traffic, not counted)
            b = 5 * 3 * H * W * 4
            out[f"l1_dssim_fwd_bwd_{W}x{H}"] = {"us": round(ms * 1e3, 1), "algorithmic_bytes": b,
                                               "frac": round(b / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                                               "us_through_the_autograd_function": round(ms_py * 1e3, 1),
                                               "what": "lr_l1_dssim_forward + lr_l1_dssim_backward, 200 back-to-back pairs on one stream "
                                                       "(at 512^2 the pair is shorter than the host takes to issue it)"}
            del gt, img, x, grad, ws
    except Exception as e:
        out["l1_dssim_error"] = str(e)[:200]
    try:
        from simple_knn._C import distCUDA2
        for P in (1_000_000, 3_000_000):
            pts = torch.rand(P, 3, device=dev, generator=None) * 4.0 - 2.0
            ms = event_ms(lambda: distCUDA2(pts), 5, warm=1)
            out[f"dist2_{P // 1_000_000}M_points"] = {"ms": round(ms, 3), "what": "Morton order + 3-NN mean squared distance, lr_dist2"}
            del pts
    except Exception as e:
        out["dist2_error"] = str(e)[:200]
    try:
        import torch.nn as nn
        from luciddreamer_amd import densify as D
        from luciddreamer_amd.optim import FusedAdam
        P = 3_000_000

        class M:
            pass
        m = M()
        mk = lambda *sh: nn.Parameter(torch.randn(*sh, device=dev).requires_grad_(True))
        m._xyz, m._features_dc, m._features_rest = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3)
        m._opacity, m._scaling, m._rotation = mk(P, 1), mk(P, 3), mk(P, 4)
        m.percent_dense = 0.01
        m.optimizer = FusedAdam([{"params": [getattr(m, a)], "lr": 1e-3, "name": n} for n, a in D.GROUP_ATTR.items()], lr=0.0, eps=1e-15)
        for n_, a in D.GROUP_ATTR.items():
            t_ = getattr(m, a)
            t_.grad = torch.zeros_like(t_) if n_ == "f_rest" else torch.randn_like(t_) * 1e-3
        # first as LucidDreamer's first thousand iterations see it: active SH degree 0, the 45 coefficients of features_rest
        # have zero gradient and zero moments -- lr_adam_step reads them (16 B) and stores nothing; then with every gradient
        # non-zero (28 B per element)
        ms0 = event_ms(lambda: m.optimizer.step(), 10)
        b0 = P * (14 * 7 + 45 * 4) * 4
        out["adam_step_3M_active_sh_degree_0"] = {"us": round(ms0 * 1e3, 1), "algorithmic_bytes": b0,
                                                  "frac": round(b0 / (ms0 * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
        for a in D.GROUP_ATTR.values():
            getattr(m, a).grad = torch.randn_like(getattr(m, a)) * 1e-3
        ms = event_ms(lambda: m.optimizer.step(), 10)
        b = P * 59 * 4 * 7                                      # read parameter, gradient, two moments; write parameter, two moments
        out["adam_step_3M"] = {"us": round(ms * 1e3, 1), "algorithmic_bytes": b, "frac": round(b / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
        m.xyz_gradient_accum, m.denom = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
        m.max_radii2D = torch.zeros(P, device=dev)
        D._store(m)
        ts = []
        for _ in range(4):
            mask = torch.rand(m._xyz.shape[0], device=dev) < 0.05
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D.prune_points(m, mask)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        rows = m._xyz.shape[0]
        b = rows * (59 * 3 + 3) * 4 * 2                         # every surviving row of the 21 tensors read and written once
        out["prune_5pct_of_3M"] = {"ms": round(min(ts[1:]) * 1e3, 3), "algorithmic_bytes": b,
                                   "frac": round(b / min(ts[1:]) / (HBM_PEAK_GBS * 1e9), 4),
                                   "what": "densify.prune_points: lr_select_rows over 6 parameters + 12 Adam moments + 3 statistics (wall clock, one call)"}
        del m
        torch.cuda.empty_cache()
    except Exception as e:
        out["densify_error"] = str(e)[:200]
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command line under torch.distributed.run, one process per
    GPU of this node, backend nccl (= RCCL).  Fails loudly -- no N = 1 fallback -- when the node shows fewer than N devices
    (LR_DIST_BACKEND=gloo lets the ranks share devices: the 1-GPU debugging configuration of tests/test_gpu_distributed.py).
    Returns the exit code of the launcher."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    backend = os.environ.get("LR_DIST_BACKEND") or "nccl"
    if have < n and backend == "nccl":
        print(f"[bench] --gpus {n} needs {n} HIP devices on this node, {have} visible: RCCL does not run two ranks on one "
              f"device.  Not falling back to fewer ranks.  (LR_DIST_BACKEND=gloo shares devices between ranks for debugging.)",
              file=sys.stderr)
        return 2
    with socket.socket() as sock:                       # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] starting {n} ranks: {' '.join(cmd[1:8])} ... (backend {backend})", file=sys.stderr)
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--no-extras", action="store_true", help="skip the entry_points / other_workloads legs (and the live PMC passes)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two rocprofv3 byte-counter passes; roofline.traffic then comes from profiles/pmc_c3.json")
    ap.add_argument("--resolution", default=None, help="WxH override (the metric uses 1920x1080)")
    ap.add_argument("--sh-degree", type=int, default=3, choices=[0, 1, 2, 3], help="active SH degree (diagnostics; the metric uses 3)")
    ap.add_argument("--api", default="views", choices=["views", "autograd", "views-loss"],
                    help="views: one lr_views_accumulate call per step (parallel.ViewBatch); autograd: the drop-in "
                         "GaussianRasterizer autograd op per view (parallel.ViewStreams); views-loss: the views step with "
                         "the fused L1+DSSIM loss against a target image formed inside (not the metric: extra work)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the views of a step alternate on (forward of view i+1 overlaps backward of view i)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1: strong = the workload's views per STEP shared by the ranks (BASELINE.json config 3; auto picks it), "
                         "weak = that many views per rank")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "sparse-rows", "sharded-adam", "split-sharded-adam"],
                    help="views API: allreduce = the metric's step (gradients all-reduced, one dense bucket); sparse-rows = the same "
                         "sum with the reduce-scatter half replaced by an all-to-all of the rows this rank's views touched "
                         "(parallel.sparse_rows_all_reduce); sharded-adam = a training step, reduce-scatter + Adam on the rank's "
                         "shard + all-gather of the parameters (extra work: not the metric)")
    ap.add_argument("--sustain-seconds", type=float, default=1.0,
                    help="after the K contract steps, time the same step for at least this long (reported as `sustained`)")
    ap.add_argument("--no-fused-accumulate", action="store_true",
                    help="let autograd accumulate dense per-view gradients instead of the in-kernel accumulation")
    args = ap.parse_args()

    from luciddreamer_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    forced = os.environ.get("LR_TUNE")                      # "knob=value,...": counter passes of a candidate kernel (tools/pmc_run.sh)
    if forced:
        from luciddreamer_amd import _lib
        for kv in forced.split(","):
            _lib.tune_set(kv.split("=")[0].strip(), int(kv.split("=")[1]))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))            # plain `python bench.py --gpus N`: start the N ranks ourselves
    rank, world, dev = parallel.init_distributed()
    if world != args.gpus:
        # never a silent N = 1 line under --gpus N: the launcher's world size and the flag must agree
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` alone: it starts the ranks itself)")
    backend = torch.distributed.get_backend() if world > 1 else None
    backend_world = torch.distributed.get_world_size() if world > 1 else 1
    collective_check = None
    if world > 1:
        # proof that the collective ran over `world` ranks on the backend named in the line: an all-reduce of ones
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        collective_check = {"all_reduce_of_ones": float(ones.item()), "ranks": backend_world, "backend": backend,
                            "devices_visible": torch.cuda.device_count()}
        if int(round(collective_check["all_reduce_of_ones"])) != args.gpus:
            raise SystemExit(f"[bench] all-reduce of ones over the ranks gave {collective_check['all_reduce_of_ones']}, not {args.gpus}")

    res = tuple(int(v) for v in args.resolution.lower().split("x")) if args.resolution else None
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    wl = Workload(args.workload, args, rank, world, dev, args.gaussians, args.views, res, scaling)
    cfg = {"workload": wl.label, "views_per_step": wl.total_views, "views_per_rank_per_step": wl.V, "gaussians": wl.P,
           "visible_mean": round(wl.V_mean, 1),
           "num_rendered_mean": round(wl.R_mean, 1), "sh_degree": args.sh_degree, "resolution": [wl.W, wl.H],
           "grad_bucket_bytes": int(wl.grads.flat.numel() * 4)}
    fused = not args.no_fused_accumulate
    value, ms_per_step, host_issue_ms, step_fn = run_leg(wl, args.api, args.exact, args.streams, args.steps, args.warmup, world, dev, fused)
    chunks = wl.chunks
    exchange_phases = getattr(wl, "exchange_phases", None)
    exchange_bytes, exchange_detail = None, None
    info = getattr(wl, "exchange_info", None)
    if info:                                            # sparse-rows: measured per step (the last `steps` calls are the timed ones)
        tail = info[-args.steps:]
        mean = lambda k: int(sum(i[k] for i in tail) / len(tail))
        exchange_bytes = mean("bytes_sent")
        exchange_detail = {"rows_sent_per_step": mean("sent_rows"), "bytes_all_to_all": mean("bytes_all_to_all"),
                           "bytes_all_gather": mean("bytes_all_gather"), "dense_ring_allreduce_bytes": mean("dense_equivalent_bytes")}
    # the contract's K steps are 0.1 s of GPU time at C3; the same step again for >= sustain-seconds as a cross-check
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(math.ceil(args.sustain_seconds / max(ms_per_step * 1e-3, 1e-6))))
        dt_sus, _ = timed(step_fn, n_sus, world, dev)
        wl.finish()
        sustained = {"views_per_s": round(wl.total_views * n_sus / dt_sus, 2), "steps": n_sus, "seconds": round(dt_sus, 3)}
    roofline = roofline_leg(wl, args.api, args.exact, args.steps, world, dev, value, args)

    extras = rank == 0 and world == 1 and not args.no_extras and args.workload == "c3" and args.api == "views" \
        and not args.exact and args.gaussians is None and args.views is None and res is None
    entry_points, other = None, None
    if extras:
        entry_points = {}
        for key, api, exact in (("drop_in_views_per_s", "autograd", False), ("exact_mode_views_per_s", "autograd", True),
                                ("views_loss_views_per_s", "views-loss", False)):
            if hasattr(wl, "pipe"):
                del wl.pipe
            v, ms, host, _ = run_leg(wl, api, exact, args.streams, args.steps, args.warmup, world, dev, fused)
            entry_points[key] = round(v, 1)
            entry_points[key.replace("_views_per_s", "_host_issue_ms_per_step")] = round(host, 3)
            if getattr(wl, "pipe", None) is not None:
                # of that, the time end_step() spent BLOCKED on the last view's header (policy "recover": the GPU is still working;
                # the rest is host work: Python, binding, launches) -- over the warm-up and timed steps of the leg
                waited = wl.pipe.waited_s / max(args.steps + args.warmup, 1) * 1e3
                entry_points[key.replace("_views_per_s", "_host_blocked_on_last_header_ms_per_step")] = round(waited, 3)
                entry_points[key.replace("_views_per_s", "_host_work_ms_per_step")] = round(max(host - waited, 0.0), 3)
        # the headline step in STRICT mode (config.set_strict_parity: alpha from the reference's own float operations -- images
        # within 1e-5 on every pixel, no gradient row exempt: `parity.strict_mode`)
        from luciddreamer_amd import config as _cfg
        _cfg.set_strict_parity(True)
        try:
            v, ms, host, _ = run_leg(wl, "views", False, args.streams, args.steps, args.warmup, world, dev, fused)
        finally:
            _cfg.set_strict_parity(False)
        entry_points["strict_mode_views_per_s"] = round(v, 1)
        # the forward-only path of the video renderer (R/luciddreamer.py:250-255: render() per frame, grad mode on, no
        # backward), default configuration, frames left on the device
        from luciddreamer_amd import config
        config.reset()
        config.set_async(True)

        def frames():
            for r in wl.rasterizers:
                r(means3D=wl.leaf["means3D"], means2D=wl.means2D, opacities=wl.leaf["opacities"], shs=wl.leaf["shs"],
                  scales=wl.leaf["scales"], rotations=wl.leaf["rotations"])
        for _ in range(args.warmup):
            frames()
        dt, host = timed(frames, args.steps, world, dev)
        entry_points["render_only_views_per_s"] = round(wl.total_views * args.steps / dt, 1)
        entry_points["render_only_host_issue_ms_per_step"] = round(host, 3)
        config.reset()
        entry_points["note"] = ("strict_mode: the headline step with the blend evaluated in the reference's own float operations; drop_in: GaussianRasterizer autograd op per view (the reference's API) with the library's "
                                f"default configuration (no config call), views pipelined over {args.streams} streams "
                                "(parallel.ViewStreams); exact_mode: the same with the reference's host round trip per view; "
                                "views_loss: the headline step with the fused L1+DSSIM loss formed inside; render_only: forward "
                                "only, one stream, default configuration (the video renderer's loop)")
    f_rows = measure_f_rows(dev) if extras else None
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity = run_cpu_baseline(wl)
    if extras:
        other = {}
        del wl
        torch.cuda.empty_cache()
        for name in ("c2", "c3box", "c4shape", "c5shape", "ld512"):
            w2 = Workload(name, args, rank, world, dev)
            steps2 = max(2, min(args.steps, 5)) * (4 if name == "c2" else 1)
            v, ms, host, _ = run_leg(w2, "views", False, args.streams, steps2, 1, world, dev, fused)
            # the stages of one view alone on the GPU (single stream, the shapes the multi-stream leg launches)
            st2, _ = profiled_stages(w2, "views", False, 2, world, dev, args, args.streams)
            per_stage = {}
            for stg in ("preprocess", "render_fwd", "render_bwd", "gauss_bwd"):
                if st2[stg][1]:
                    ab = stage_bytes_moved(stg, w2.P, w2.V_mean, w2.R_mean, w2.N, w2.T, w2.K, w2.M)
                    us = st2[stg][0] / st2[stg][1] * 1e3
                    per_stage[stg] = {"us": round(us, 1), "frac": round(ab / (us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)}
            b_f, b_b = path_bytes(w2.P, w2.V_mean, w2.R_mean, w2.N, w2.T, w2.K, w2.M)
            b_m = moved_bytes(w2.P, w2.V_mean, w2.R_mean, w2.N, w2.T, w2.K, w2.M, w2.V)
            other[name] = {"workload": w2.label, "value": round(v, 1), "unit": "views/s", "steps": steps2,
                           "ms_per_step": round(ms, 3), "visible_mean": round(w2.V_mean, 1),
                           "num_rendered_mean": round(w2.R_mean, 1),
                           "path_frac_moved": round(b_m * v / (HBM_PEAK_GBS * 1e9), 5),
                           "single_view_stage_us_and_frac_of_hbm_peak": per_stage}
            del w2
            torch.cuda.empty_cache()

    if rank == 0:
        from luciddreamer_amd import _lib
        metric = {"c2": "views/sec fwd+bwd @1080p (1e5 Gaussians)", "c4shape": "views/sec fwd+bwd @1440p (3e6 Gaussians)",
                  "c5shape": "views/sec fwd+bwd @512x512 (1e6 Gaussians)"}.get(args.workload, "views/sec fwd+bwd @1080p (1e6 Gaussians)")
        bucket_bytes = cfg["grad_bucket_bytes"]
        cfg.update({"mode": "exact (host sync per view)" if args.exact else "async (no host sync per view)",
                    "parallelism": (f"dp{world} (view i -> rank i mod {world}; gradients all-reduced "
                                    + ("once per step" if chunks == 1 else f"in {chunks} view groups, the first under the second")
                                    + ")") if world > 1 else "dp1",
                    # bytes every rank puts on its links per step (ring model: a dense fp32 all-reduce of L floats sends
                    # 2 (n - 1) / n x 4 L; sparse-rows: measured rows x (4 + row bytes) + the all-gather half)
                    "allreduce_bytes_per_step": (exchange_bytes if exchange_bytes is not None else
                                                 int(2 * (world - 1) * bucket_bytes * chunks // world)) if world > 1 else 0,
                    "allreduce_payload_bytes_per_step": int(bucket_bytes * chunks) if world > 1 else 0,
                    "exchange_detail": exchange_detail if exchange_detail is not None else exchange_phases,
                    "dist_backend": backend, "dist_world_size": backend_world, "collective_check": collective_check,
                    "streams_per_rank": args.streams,
                    "view_chains": (f"{args.streams} chains of views in flight: the last view's chain on the caller's stream, "
                                    f"{max(0, args.streams - 1)} on streams of the library (csrc/api.hip views_core)"
                                    if args.api in ("views", "views-loss") else f"parallel.ViewStreams({args.streams})"),
                    "api": args.api, "exchange": args.exchange, "host_issue_ms_per_step": round(host_issue_ms, 3),
                    "lr_version": _lib.lib().lr_version().decode()})
        if os.environ.get("LR_TUNE"):
            cfg["forced_kernel_variants"] = os.environ["LR_TUNE"]
        # the figures beside the headline that a reader of the driver's record needs (the driver keeps `config` whole):
        # the same step in strict mode, the API LucidDreamer calls, the >= 1 s repeat of the headline step, LucidDreamer's own
        # loop after install(), and whether the headline's own kernels met the oracle at this size
        c5 = (cpu_baseline or {}).get("c5_train_loop") or {}
        hs = (parity or {}).get("headline_step") if isinstance(parity, dict) else None
        cfg.update({
            "strict_mode_views_per_s": (entry_points or {}).get("strict_mode_views_per_s"),
            "drop_in_views_per_s": (entry_points or {}).get("drop_in_views_per_s"),
            "drop_in_host_issue_ms_per_step": (entry_points or {}).get("drop_in_host_issue_ms_per_step"),
            "drop_in_host_work_ms_per_step": (entry_points or {}).get("drop_in_host_work_ms_per_step"),
            "sustained_views_per_s": (sustained or {}).get("views_per_s"),
            "c5_train_loop_ms_per_iter": {k: (c5.get(k) or {}).get("ms_per_iter") for k in
                                          ("this_rasterizer", "this_rasterizer_after_install", "this_rasterizer_after_install_fuse_step",
                                           "with_optional_pieces",
                                           "reference_kernels_on_this_gpu")} if c5 and "error" not in c5 else c5.get("error"),
            "headline_step_parity": None if not hs else {
                m: {"kernel_shapes": hs[m]["kernel_shapes"], "grad_rows_above_1e-4_worst_tensor": hs[m]["grad_rows_above_1e-4_worst_tensor"],
                    "worst_grad_row_rel": hs[m]["worst_grad_row_rel"],
                    "every_row_above_1e-4_touches_a_flagged_pixel": hs[m]["every_row_above_1e-4_touches_a_flagged_pixel"]}
                for m in ("default", "strict")},
            "headline_kernels_image_parity": None if not isinstance(parity, dict) or "per_view" not in parity else {
                "max_abs_rgb_err": max(v["headline_kernels"]["max_abs_rgb_err"] for v in parity["per_view"]),
                "pixels_above_1e-5_unmasked": max(v["headline_kernels"]["pixels_above_1e-5_unmasked"] for v in parity["per_view"]),
                "kernel_shapes": parity["per_view"][0]["headline_kernels"]["kernel_shapes"]},
        })
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # the same step with the blend evaluated in the reference's own float operations (config.set_strict_parity): the
            # figure that meets the stated tolerance on EVERY pixel and gradient row with nothing masked (parity.strict_mode)
            "strict_mode_views_per_s": (entry_points or {}).get("strict_mode_views_per_s"),
            "strict_mode_parity": (parity or {}).get("strict_mode") if isinstance(parity, dict) else None,
            "config": cfg,
            "roofline": roofline,
            "parity": parity,
            "sustained": sustained,
            "entry_points": entry_points,
            "other_workloads": other,
            "f_rows": f_rows,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def _hip_one_view(wl, cam_index=0, views_in_flight=1):
    """One view of the workload through the drop-in operator in exact mode, with its own leaves (dense gradients of that
    view alone): what `parity` compares with the oracle.  Outside every timed region.  views_in_flight >= 2: the hint the
    multi-stream entry points give the library, so that the kernels of the HEADLINE run (k_render_fwd_tile / k_render_bwd_tile at
    1080p) and not the lone-view shapes; the shapes that ran are returned."""
    from luciddreamer_amd import _lib, config
    config.reset()
    config.set_async(False)
    config.set_fused_grad_accumulation(False)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in wl.leaf.items()}
    means2D = torch.zeros(wl.P, 3, device=wl.dev, requires_grad=True)
    _lib.tune_set("views_in_flight", views_in_flight if views_in_flight >= 2 else -1)
    try:
        color, radii, depth = wl.rasterizers[cam_index](means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"],
                                                         shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
        color.backward(wl.grad_color)
        torch.cuda.synchronize()
        shapes = _lib.last_launch_shapes()
    finally:
        _lib.tune_set("views_in_flight", -1)
    g = lambda t: t.grad.detach().cpu().numpy()
    return {"color": color.detach().cpu().numpy(), "depth": depth.detach().cpu().numpy(), "radii": radii.cpu().numpy(),
            "kernel_shapes": {"forward": shapes[0], "backward": shapes[1]},
            "grads": {"means2D": g(means2D), "opacity": g(leaf["opacities"]), "means3D": g(leaf["means3D"]),
                      "sh": g(leaf["shs"]), "scales": g(leaf["scales"]), "rotations": g(leaf["rotations"])}}


def _hip_headline_step(wl, view_ids, n_streams=3):
    """The HEADLINE's own entry point and kernels on a few views: ONE lr_views_accumulate call (parallel.ViewBatch, async mode,
    `n_streams` views in flight -- at 1080p k_render_fwd_tile + k_render_bwd_tile) accumulating the views' gradients into
    zeroed tensors.  Returns the accumulated gradients (numpy) and the kernel shapes that ran.  Outside every timed region."""
    from luciddreamer_amd import _lib, config, parallel
    config.reset()
    config.set_async(True)
    cams = [wl.cams[i] for i in view_ids]
    batch = parallel.ViewBatch(cams, [wl.grad_color] * len(cams), wl.degree, wl.bg, wl.capacity, n_streams=n_streams)
    leaf = wl.leaf
    acc = {"means3D": torch.zeros_like(leaf["means3D"]), "means2D": torch.zeros(wl.P, 3, device=wl.dev),
           "opacity": torch.zeros_like(leaf["opacities"]), "sh": torch.zeros_like(leaf["shs"]),
           "scales": torch.zeros_like(leaf["scales"]), "rotations": torch.zeros_like(leaf["rotations"])}
    with torch.no_grad():
        batch.run(leaf["means3D"].detach(), leaf["opacities"].detach(), leaf["scales"].detach(), leaf["rotations"].detach(),
                  leaf["shs"].detach(), acc)
    batch.check()                                   # synchronises; raises if a view overflowed its binning capacity
    shapes = _lib.last_launch_shapes()
    return {k: v.cpu().numpy() for k, v in acc.items()}, {"forward": shapes[0], "backward": shapes[1]}


def run_cpu_baseline(wl):
    """The reported CPU baseline and the accuracy half of the metric, on rank 0 at N = 1, outside every timed region.

    Baseline: the reference's OWN rasterizer sources compiled for the host (oracle/_ref/libref_raster.so, "reference")
    when the build container shipped it, else the restatement ("port", pinned to it bit for bit by
    tests/test_oracle_ref.py), forward + backward of views of the same workload on all host cores, bounded to ~20 s;
    the other of the two is reported beside it.  Parity: view 0 of the workload rendered by the HIP path against the
    restatement's image, depth, radii and gradients (max-abs errors; the oracle flags the pixels that sit within
    rounding of one of the reference's discrete thresholds)."""
    import numpy as np
    from luciddreamer_amd import synthetic
    from oracle import oracle
    cloud, cams, degree, H, W = wl.cloud, wl.my_cams, wl.degree, wl.H, wl.W
    n = lambda t: t.detach().cpu().numpy()
    g = n(synthetic.upstream_grad(H, W))
    oracle.lib()

    def one_view(mod, c, keep=False):
        tfx, tfy = math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5)
        t0 = time.perf_counter()
        res = mod.forward(np.zeros(3, np.float32), n(cloud["means3D"]), None, n(cloud["opacities"]), n(cloud["scales"]),
                          n(cloud["rotations"]), 1.0, None, n(c.world_view_transform), n(c.full_proj_transform),
                          tfx, tfy, H, W, n(cloud["shs"]), degree, n(c.camera_center))
        t1 = time.perf_counter()
        grads = mod.backward(res, g)
        t2 = time.perf_counter()
        return (t1 - t0, t2 - t1) + ((res, grads) if keep else ())

    def timed_views(mod, budget_s):
        one_view(mod, cams[0])                              # warm-up (thread pool, page faults)
        times, t_start = [], time.perf_counter()
        for i in range(10):                                 # up to 10 views of the path, bounded
            times.append(one_view(mod, cams[i % len(cams)])[:2])
            if time.perf_counter() - t_start > budget_s:
                break
        tot = sorted(f + b for f, b in times)
        med = tot[len(tot) // 2]
        fwd_med = sorted(f for f, _ in times)[len(times) // 2]
        return {"value": round(1.0 / med, 4), "unit": "views/s",
                "sample": f"median of {len(times)} views fwd+bwd of the same workload after 1 warm-up view "
                          f"({fwd_med:.2f}s fwd + {med - fwd_med:.2f}s bwd)"}

    def reference_train_loop(iters=60):
        """BASELINE.json config 5's shape as the reference runs it: the UNCHANGED LucidDreamer training iteration
        (R/luciddreamer.py:283-327 around the reference's own GaussianModel / render() / loss, tests/ref_loop.py) at 1 M
        Gaussians, 512 x 512, batch 1 -- once over this repository's rasterizer (the drop-in, nothing configured) and once over
        the reference's own kernels compiled for this GPU (oracle/_ref, the stated baseline).  Baseline-leg infrastructure:
        the reference's Python and its kernels come from oracle/_ref; the product under the loop is the HIP rasterizer."""
        try:
            import numpy as np
            from luciddreamer_amd import cameras, config
            from tests import ref_loop
            from tests.test_gpu_reference_stack import _perturbed, _targets
            from oracle import ref_device, ref_python
            if not ref_python.available():
                return {"skipped": "the reference's Python layer is not staged (oracle/_ref/py)"}
            config.reset()
            config.set_async(True)                              # the library's defaults: what an unchanged caller gets
            config.set_fused_grad_accumulation(False)
            P, W, H = 1_000_000, 512, 512
            cams = cameras.lookaround_path(W, H, n_views=8, max_yaw_deg=8.0, max_pitch_deg=4.0)
            base, hidden = _perturbed(P, 41)
            targets, depths = _targets(hidden, cams)
            order = [int(i) for i in np.random.default_rng(9).integers(0, 8, size=iters)]
            out = {"workload": f"unchanged reference training iteration (render -> L1+DSSIM -> backward -> Adam), {P} Gaussians, "
                               f"{W}x{H}, batch 1, {iters} iterations after a {iters}-iteration warm-up pass; cameras, targets and optimizer in place "
                               f"before the clock starts, as in the reference"}
            backends = ["ours"] + (["refdev"] if ref_device.available() else [])
            for be in backends + backends:                        # first pass of each: warm-up (MIOpen tuning, allocator)
                with ref_loop.stack(be) as (R, dev):
                    gm = ref_loop.model_from_cloud(R, base, dev)
                    # cameras and targets on the device, optimizer built: as when the reference enters its loop (the timed region
                    # is iterations only; through round 4's first runs it also held the harness's own host -> device copies
                    # of the 16 target images, ~1.2 ms per iteration at 60 iterations, in every variant of the unchanged loop)
                    cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    res = ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                key = "this_rasterizer" if be == "ours" else "reference_kernels_on_this_gpu"
                out[key] = {"iter_per_s": round(iters / dt, 1), "ms_per_iter": round(dt / iters * 1e3, 3),
                            "final_loss": round(float(res["loss"][-1]), 5)}
            # the UNCHANGED loop again after ONE call, luciddreamer_amd.install(R): render_raw, the paired l1 / ssim pass,
            # FusedAdam and the fused densification statistics are switched in underneath the reference's own names
            import luciddreamer_amd
            for policy, key, what, active in (
                    ("verify", "this_rasterizer_after_install",
                     "the same unchanged loop after luciddreamer_amd.install(reference modules): one call, no edits to the caller", None),
                    ("verify", "this_rasterizer_after_install_fuse_step",
                     "the same after install(reference modules, fuse_step=True): on every plain iteration the backward hands its gradients "
                     "to the optimizer privately -- visited rows only, no zero-fill (LR_ACC_NO_ZERO_FILL) -- and launches the masked "
                     "Adam step (lr_adam_step_masked) right behind its own kernels; optimizer.step() only checks the iteration -- "
                     "parameters bit-identical to the line above "
                     "(tests/test_gpu_optim.py, tests/test_gpu_reference_stack.py)", None),
                    ("drop", "this_rasterizer_after_install_policy_drop",
                     "the same with config.set_async(True, on_overflow='drop'): the forward does not wait for its own header (a view "
                     "that needs more than 1.3 x the instances of any view before it is warned about and contributes no gradient)", None),
                    ("verify", "this_rasterizer_after_install_active_sh_degree_0",
                     "the installed loop as LucidDreamer's FIRST thousand iterations run it: active SH degree 0 of 3 (R/luciddreamer.py:"
                     "287-288 raises it every 1000 of 2990 iterations; every leg above runs degree 3 from the start) -- the 45 "
                     "coefficients of features_rest receive no gradient and the Adam step stores nothing for them", 0)):
                config.reset()
                config.set_async(True, on_overflow=policy)
                for _pass in range(2):
                    with ref_loop.stack("ours") as (R, dev):
                        handle = luciddreamer_amd.install(R, backward_on_calling_thread=True,    # the single-threaded loop: what "auto" picks there
                                                          fuse_step=key.endswith("_fuse_step"))
                        try:
                            gm = ref_loop.model_from_cloud(R, base, dev, active_sh_degree=active)
                            cams_r, tg_r, dg_r, opt_r = ref_loop.resident(R, gm, dev, cams, targets, depths, iters)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            res = ref_loop.train(R, gm, dev, cams_r, order, tg_r, dg_r, iters=iters, opt=opt_r)
                            torch.cuda.synchronize()
                            dt = time.perf_counter() - t0
                        finally:
                            luciddreamer_amd.uninstall(handle)
                out[key] = {"iter_per_s": round(iters / dt, 1), "ms_per_iter": round(dt / iters * 1e3, 3),
                            "final_loss": round(float(res["loss"][-1]), 5), "what": what}
            config.reset()
            config.set_async(True)
            # the same iteration with the optional pieces of SURVEY.md section 8f switched in (one line
            # each in the reference's loop): render_raw (activations inside the kernels), the fused L1+DSSIM loss, the
            # fused Adam step and densification statistics -- same cloud, cameras, targets, view order and loss terms
            import importlib.util
            spec = importlib.util.spec_from_file_location("lr_example_train_loop", os.path.join(ROOT, "examples", "train_loop.py"))
            ex = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ex)
            from luciddreamer_amd import densify
            from luciddreamer_amd.gaussian_renderer import render_raw
            from luciddreamer_amd.loss import l1_dssim_loss
            dev = torch.device("cuda:0")
            cams_d = [c.to(dev) for c in cams]
            tg, dg = [t.to(dev) for t in targets], [t.to(dev) for t in depths]
            bg = torch.zeros(3, device=dev)
            for _pass in range(2):                                # first pass: warm-up, as above
                b = {k: v.to(dev) for k, v in base.items()}
                model = ex.TrainableCloud(b["means3D"], b["scales"], b["rotations"], b["opacities"], b["shs"])
                model.training_setup({"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3,
                                      "rotation": 1e-3})
                import gc
                gc.collect()                                      # as ref_loop.resident(): the set-up's objects are old now
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for it in range(iters):
                    k = order[it]
                    pkg = render_raw(cams_d[k], model, bg_color=bg)
                    loss = l1_dssim_loss(pkg["render"], tg[k], 0.2) + 0.1 * (pkg["depth"] - dg[k]).abs().mean()
                    loss.backward()
                    with torch.no_grad():
                        densify.add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
                        model.optimizer.step()
                        model.optimizer.zero_grad(set_to_none=True)
                    last_t = loss.detach()                        # like the reference's loop: the loss is not read inside
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                last = float(last_t)
            out["with_optional_pieces"] = {"iter_per_s": round(iters / dt, 1), "ms_per_iter": round(dt / iters * 1e3, 3),
                                           "final_loss": round(last, 5),
                                           "what": "render_raw + l1_dssim_loss + FusedAdam + add_densification_stats "
                                                   "(luciddreamer_amd, SURVEY.md 8f) in place of render / l1+ssim / "
                                                   "torch.optim.Adam / the mask-indexed statistics"}
            return out
        except Exception as e:
            return {"error": str(e)[:300], "trace": traceback.format_exc()[-600:]}

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
    port = timed_views(oracle, 12.0)
    port["what"] = "oracle/raster_oracle.c (restatement of the reference semantics), OpenMP"
    out = dict(port, cores=cores, kind="port", cpu_model=model)
    out["sample"] += f", OpenMP over {cores} host threads"
    try:
        from oracle import ref
        if ref.available():
            ref.set_threads(0)
            r = timed_views(ref, 15.0)
            r["what"] = ("the reference's forward.cu / backward.cu / rasterizer_impl.cu compiled for the host by "
                         "oracle/build_ref.py (blocks over OpenMP threads, threads of a block as fibers)")
            out = dict(r, cores=cores, kind="reference", cpu_model=model, port=port)
            out["sample"] += f", {cores} host threads"
    except Exception as e:                                   # optional evidence, never a reason to fail
        out["reference_host_error"] = str(e)[:200]

    # ---- accuracy of the HIP path against the oracle (the third part of BASELINE.json's metric): views 0 / 11 / 19 of the
    # path (as many as the workload has), every figure the worst over the views
    parity = None
    try:
        names = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
        per_view = []
        view_ids = [i for i in (0, 11, 19) if i < len(cams)] or [0]
        oracle_sum = {}                 # sum over the views of the oracle's gradients (float64): what the headline's step accumulates
        touch_info = []                 # per view: what decides whether a row beyond 1e-4 sits on an oracle-flagged pixel

        def touches_flagged(i, st, fx, fy):
            """Row i belongs to a splat whose own alpha reaches the 1/255 threshold (to within 10 %) on a pixel the oracle
            flags as sitting within float32 rounding of a discrete decision."""
            ca, cb, cc, op = st["conic_opacity"][i].astype(np.float64)
            dx, dy = st["means2D"][i, 0] - fx.astype(np.float64), st["means2D"][i, 1] - fy.astype(np.float64)
            power = -0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy
            return bool(((power <= 1e-6) & (op * np.exp(np.minimum(power, 0.0)) >= 0.9 / 255.0)).any())

        for vi in view_ids:
            _, _, res, grads = one_view(oracle, cams[vi], keep=True)
            hip = _hip_one_view(wl, vi)
            st = res.stage()
            frag = st["fragile"]
            fc, fd = (frag & 1) != 0, (frag & 2) != 0
            fy, fx = np.nonzero(frag != 0)
            cerr = np.abs(hip["color"] - res.color)
            derr = np.abs(hip["depth"][0] - res.depth[0]) / np.maximum(1.0, np.abs(res.depth[0]))
            ref_g = dict(zip(names, grads[:8]))
            for k in hip["grads"]:
                b64 = ref_g[k].reshape(hip["grads"][k].shape).astype(np.float64)
                oracle_sum[k] = b64 if k not in oracle_sum else oracle_sum[k] + b64
            touch_info.append({"conic_opacity": st["conic_opacity"].copy(), "means2D": st["means2D"].copy(), "fx": fx, "fy": fy})
            gerr, all_touch = {}, True
            for k, a in hip["grads"].items():
                b = ref_g[k].reshape(a.shape)
                scale = float(np.abs(b).max())
                row = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
                bad = np.nonzero(row > 1e-4 * scale)[0]
                # a row beyond the tolerance must belong to a splat whose own alpha reaches the 1/255 threshold (to within
                # 10 %) on a pixel the oracle flags as sitting within float32 rounding of a discrete decision
                for i in bad:
                    all_touch &= touches_flagged(i, st, fx, fy)
                gerr[k] = {"max_rel": float(row.max() / scale) if scale > 0 else float(row.max()), "rows_above_1e-4": int(len(bad))}
            # the same view through the kernels the HEADLINE launches (three views in flight: at 1080p the one-wave-per-tile pair)
            ht = _hip_one_view(wl, vi, views_in_flight=3)
            t_cerr = np.abs(ht["color"] - res.color)
            t_rows, t_worst, t_touch = 0, 0.0, True
            for k, a in ht["grads"].items():
                b = ref_g[k].reshape(a.shape)
                scale = float(np.abs(b).max())
                row = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
                bad = np.nonzero(row > 1e-4 * scale)[0]
                t_rows = max(t_rows, int(len(bad)))
                t_worst = max(t_worst, float(row.max() / scale) if scale > 0 else 0.0)
                for i in bad:
                    t_touch &= touches_flagged(i, st, fx, fy)
            headline_kernels_view = {
                "kernel_shapes": ht["kernel_shapes"], "max_abs_rgb_err": float(t_cerr[:, ~fc].max()),
                "pixels_above_1e-5_unmasked": int((t_cerr.max(axis=0) > 1e-5).sum()),
                "same_bits_as_lone_view_kernels": bool(np.array_equal(ht["color"], hip["color"]) and np.array_equal(ht["depth"], hip["depth"])),
                "radii_exact": bool(np.array_equal(ht["radii"], res.radii)),
                "grad_rows_above_1e-4_worst_tensor": t_rows, "worst_grad_row_rel": t_worst,
                "every_row_above_1e-4_touches_a_flagged_pixel": t_touch}
            del ht
            # the same view in STRICT mode (config.set_strict_parity: the reference's own float operations): nothing masked
            from luciddreamer_amd import config as _cfg
            _cfg.set_strict_parity(True)
            try:
                hs = _hip_one_view(wl, vi)
            finally:
                _cfg.set_strict_parity(False)
            s_cerr = np.abs(hs["color"] - res.color)
            s_rows, s_worst = 0, 0.0
            for k, a in hs["grads"].items():
                b = ref_g[k].reshape(a.shape)
                scale = float(np.abs(b).max())
                row = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
                s_rows += int((row > 1e-4 * scale).sum())
                s_worst = max(s_worst, float(row.max() / scale) if scale > 0 else 0.0)
            strict_view = {"max_abs_rgb_err_unmasked": float(s_cerr.max()), "pixels_above_1e-5_unmasked": int((s_cerr.max(axis=0) > 1e-5).sum()),
                           "grad_rows_above_1e-4": s_rows, "worst_grad_row_rel": s_worst,
                           "radii_exact": bool(np.array_equal(hs["radii"], res.radii))}
            del hs
            per_view.append({
                "view": vi, "strict_mode": strict_view, "headline_kernels": headline_kernels_view, "max_abs_rgb_err": float(cerr[:, ~fc].max()), "max_abs_rgb_err_unmasked": float(cerr.max()),
                "max_abs_depth_rel_err": float(derr[~(fc | fd)].max()), "max_abs_depth_rel_err_unmasked": float(derr.max()),
                "threshold_pixels_flagged_by_oracle": int(fc.sum()), "pixels_above_1e-5_unmasked": int((cerr.max(axis=0) > 1e-5).sum()),
                "radii_exact": bool(np.array_equal(hip["radii"], res.radii)), "grad_err_vs_tensor_max": gerr,
                "every_row_above_1e-4_touches_a_flagged_pixel": all_touch})
            del res, grads, hip
        # ---- the headline's own entry point: one lr_views_accumulate call over the same views, three in flight, against the
        # SUM of the oracle's backwards (backward.cu:399-586 per view; gradients are additive over views), default and strict
        headline_step = {}
        for mode in ("default", "strict"):
            from luciddreamer_amd import config as _cfg
            _cfg.set_strict_parity(mode == "strict")
            try:
                got, shapes = _hip_headline_step(wl, view_ids, n_streams=3)
            finally:
                _cfg.set_strict_parity(False)
            rows_worst, rel_worst, touch, per_tensor = 0, 0.0, True, {}
            for k, a in got.items():
                b = oracle_sum[k].reshape(a.shape)
                scale = float(np.abs(b).max())
                row = np.abs(a.astype(np.float64) - b).reshape(a.shape[0], -1).max(axis=1)
                bad = np.nonzero(row > 1e-4 * scale)[0]
                for i in bad:
                    touch &= any(touches_flagged(i, t, t["fx"], t["fy"]) for t in touch_info)
                per_tensor[k] = {"max_rel": float(row.max() / scale) if scale > 0 else float(row.max()), "rows_above_1e-4": int(len(bad))}
                rows_worst, rel_worst = max(rows_worst, int(len(bad))), max(rel_worst, per_tensor[k]["max_rel"])
            headline_step[mode] = {"kernel_shapes": shapes, "grad_err_vs_summed_oracle": per_tensor,
                                   "grad_rows_above_1e-4_worst_tensor": rows_worst, "worst_grad_row_rel": rel_worst,
                                   "every_row_above_1e-4_touches_a_flagged_pixel": touch}
            del got
        headline_step["what"] = (f"ONE lr_views_accumulate call over views {view_ids} with 3 views in flight (the headline's entry point "
                                 "and kernels) vs the sum of the CPU oracle's per-view backwards; images of the same kernels: "
                                 "per_view[].headline_kernels")
        del oracle_sum, touch_info
        worst = lambda key: max(v[key] for v in per_view)
        parity = {
            "views": [v["view"] for v in per_view], "what": "drop-in operator (exact mode) vs the CPU oracle; worst over the views",
            "max_abs_rgb_err": worst("max_abs_rgb_err"), "max_abs_rgb_err_unmasked": worst("max_abs_rgb_err_unmasked"),
            "max_abs_depth_rel_err": worst("max_abs_depth_rel_err"),
            "max_abs_depth_rel_err_unmasked": worst("max_abs_depth_rel_err_unmasked"),
            "threshold_pixels_flagged_by_oracle": worst("threshold_pixels_flagged_by_oracle"), "pixels": int(H * W),
            "pixels_above_1e-5_unmasked": worst("pixels_above_1e-5_unmasked"),
            "radii_exact": all(v["radii_exact"] for v in per_view),
            "grad_err_vs_tensor_max": {k: {"max_rel": max(v["grad_err_vs_tensor_max"][k]["max_rel"] for v in per_view),
                                           "rows_above_1e-4": max(v["grad_err_vs_tensor_max"][k]["rows_above_1e-4"] for v in per_view)}
                                       for k in per_view[0]["grad_err_vs_tensor_max"]},
            "every_row_above_1e-4_touches_a_flagged_pixel": all(v["every_row_above_1e-4_touches_a_flagged_pixel"] for v in per_view),
            "tolerance": {"rgb": 1e-5, "depth_rel": 1e-5, "grad_rel": 1e-4}, "per_view": per_view,
            "headline_step": headline_step,
            # strict mode (entry_points.strict_mode_views_per_s is its throughput): worst over the views, NOTHING masked or exempt
            "strict_mode": {"max_abs_rgb_err_unmasked": max(v["strict_mode"]["max_abs_rgb_err_unmasked"] for v in per_view),
                            "pixels_above_1e-5_unmasked": max(v["strict_mode"]["pixels_above_1e-5_unmasked"] for v in per_view),
                            "grad_rows_above_1e-4": max(v["strict_mode"]["grad_rows_above_1e-4"] for v in per_view),
                            "worst_grad_row_rel": max(v["strict_mode"]["worst_grad_row_rel"] for v in per_view),
                            "radii_exact": all(v["strict_mode"]["radii_exact"] for v in per_view)},
        }
        cal = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "parity_calibration.json")
        if os.path.exists(cal):
            # the budget of those exemptions: how far the reference's own builds are from each other on the same views
            # (tests/test_gpu_ref_selfcal.py wrote the file on a GPU box; counts only, stamped with the build it was run on)
            cj = json.load(open(cal))
            pick = lambda c, pair: None if c.get(pair) is None else {"pixels_beyond_1e-5": c[pair]["pixels_beyond_1e-5"],
                                                                     "rows_beyond_1e-4": c[pair]["rows_beyond_1e-4"]}
            parity["reference_self_disagreement"] = {
                "source": "profiles/parity_calibration.json", "measured_on": cj.get("lr_version"),
                "cases": {name: {"reference_default_contraction_vs_strict": pick(c, "reference_fma_vs_reference_gfx950"),
                                 "reference_host_vs_gfx950_strict": pick(c, "reference_host_vs_reference_gfx950"),
                                 "hip_vs_reference_strict": pick(c, "hip_vs_reference_gfx950")}
                          for name, c in cj.get("cases", {}).items()}}
    except Exception as e:
        parity = {"error": str(e)[:300], "trace": traceback.format_exc()[-600:]}

    # The reference's OWN kernels (forward.cu, backward.cu, rasterizer_impl.cu compiled by hipcc for gfx950,
    # oracle/build_ref.py build_device()) on this GPU, driven the way its binding drives them: zero-filled outputs and
    # gradients, one blocking read-back per forward, legacy stream.
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
            nv = 0
            for _ in range(3):
                for x in cd:
                    ref_view(x)
                    nv += 1
            ref_device.lib().refdev_sync()
            dt = time.perf_counter() - t0
            out["reference_kernels_on_this_gpu"] = {
                "value": round(nv / dt, 1), "unit": "views/s",
                "what": "the reference's own CUDA sources compiled by hipcc for gfx950 (oracle/_ref, -O3 -ffp-contract=off, hipCUB sort/scan), "
                        f"{nv} views fwd+bwd of the same workload, one stream, its own host read-back per forward"}
    except Exception as e:                                   # the baseline is optional evidence, never a reason to fail
        out["reference_kernels_on_this_gpu"] = {"error": str(e)[:200]}
    if wl.name == "c3" and not getattr(wl, "skip_train_loop", False):
        out["c5_train_loop"] = reference_train_loop()
    return out, parity


if __name__ == "__main__":
    main()
