"""View-level data parallelism for the rasterizer: one process per GPU, views of a camera path
sharded across ranks, ONE flat fp32 gradient all-reduce per optimisation step (RCCL over xGMI via
torch.distributed backend "nccl"; "gloo" on CPU for tests).

The reference has no distributed code on this path; its training loop renders
one view per iteration (/root/reference/luciddreamer.py:291-304).  Views are independent given the
(replicated) Gaussian parameters and their gradients add, so the path shards with a single exchange:

    rank r renders views r, r+n, r+2n, ... of the step and accumulates parameter gradients locally
    into a FlatGrads bucket (the .grad of every parameter is a view into one contiguous buffer, so
    autograd accumulates in place and nothing is packed or copied for the collective);
    all_reduce(SUM) of the bucket -- 59 floats = 236 B per Gaussian (xyz 3, f_dc 3, f_rest 45,
    opacity 1, scaling 3, rotation 4; /root/reference/scene/gaussian_model.py:143-148);
    identical optimiser steps on every rank keep the replicas bit-identical.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large message per step lets RCCL use all
links; many small per-tensor all-reduces would be latency- and per-link-bound.  Densification
statistics consumed by the unchanged GaussianModel are reduced as well: xyz_gradient_accum and denom
(SUM) and max_radii2D (MAX) (scene/gaussian_model.py:405-407, luciddreamer.py:310-311).
"""
import os
from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, device).  No-op for WORLD_SIZE == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if device is None:
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            device = torch.device("cuda", local % torch.cuda.device_count())
        else:
            device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm; LR_DIST_BACKEND=gloo lets several ranks share one GPU (debugging on a 1-GPU box:
            # RCCL refuses two ranks on the same device, gloo stages the all-reduce through the host)
            backend = os.environ.get("LR_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = device
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        except TypeError:                          # a torch without the device_id keyword
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, device


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_views(n_views: int, rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    """Indices of the views rank `rank` renders: rank-strided (view i -> rank i mod n)."""
    rank = get_rank() if rank is None else rank
    world = world_size() if world is None else world
    return list(range(rank, n_views, world))


class FlatGrads:
    """One contiguous fp32 buffer holding the gradients of all parameters; p.grad are views into it."""

    ALIGN = 4      # floats: every segment starts on a 16-byte boundary (the HIP kernels accumulate with 16-byte accesses)

    def __init__(self, params: Sequence[torch.Tensor], multiple_of: int = 1):
        """multiple_of: round the bucket length up to a multiple of this many floats (ShardedAdam: world x 4, so that
        every rank's shard of a reduce-scatter starts on a 16-byte boundary)."""
        self.params = list(params)
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = sum(pad(p.numel()) for p in self.params)
        total = (total + multiple_of - 1) // multiple_of * multiple_of
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.views, self.segments = [], []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            self.segments.append((off, p.numel()))
            off += pad(p.numel())

    @classmethod
    def like(cls, other: "FlatGrads") -> "FlatGrads":
        """A second bucket with the layout of `other` that does NOT become the parameters' .grad (accumulation target of
        a group of views whose all-reduce overlaps the next group, see ChunkedViewStep)."""
        self = cls.__new__(cls)
        self.params, self.segments = [], list(other.segments)
        self.flat = torch.zeros_like(other.flat)
        self.views = [self.flat[off:off + n].view_as(v) for (off, n), v in zip(other.segments, other.views)]
        return self

    def zero_(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):   # re-attach in case an optimiser set grads to None
            p.grad = v

    def all_reduce(self, average: bool = False, async_op: bool = False):
        if world_size() == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if average and not async_op:
            self.flat.div_(world_size())
        return work


class FlatParams:
    """The parameters themselves as views of ONE contiguous fp32 buffer, laid out like a FlatGrads bucket of the same
    tensors (same 16-byte-padded segments): `p.data` of every parameter is re-pointed into the buffer (values copied
    once), so the all-gather of ShardedAdam updates every parameter in place with no packing."""

    def __init__(self, params: Sequence[torch.Tensor], like: FlatGrads):
        self.params = list(params)
        self.flat = torch.zeros_like(like.flat)
        self.views = []
        for p, (off, n) in zip(self.params, like.segments):
            v = self.flat[off:off + n].view_as(p)
            v.copy_(p.detach())
            p.data = v
            self.views.append(v)


def _adam_pieces_hip(pieces, beta1, beta2, eps, step):
    """Default arithmetic of ShardedAdam: lr_adam_step (adam.hip) over up to 16 slices per launch."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    for i in range(0, len(pieces), 16):
        chunk = pieces[i:i + 16]
        n = len(chunk)
        arr = lambda k: (ctypes.c_void_p * n)(*[it[k].data_ptr() for it in chunk])
        numel = (ctypes.c_ulonglong * n)(*[it[0].numel() for it in chunk])
        lrs = (ctypes.c_double * n)(*[float(it[4]) for it in chunk])
        dev = chunk[0][0].device
        with torch.cuda.device(dev):
            rc = L.lr_adam_step(n, arr(0), arr(1), arr(2), arr(3), numel, lrs, float(beta1), float(beta2), float(eps),
                                int(step), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "lr_adam_step")


class ShardedAdam:
    """Optimizer step of the data-parallel loop with the exchange split in two and the optimizer sharded (ZeRO-1 shape):

        reduce_scatter(SUM) of the flat gradient bucket   rank r ends up with floats [r L/n, (r+1) L/n) of the sum
        Adam on that shard                                 moments exist for the shard only
        all_gather of the flat PARAMETER buffer            every rank has every updated parameter again

    The links carry what one all-reduce carries (a ring / mesh all-reduce IS a reduce-scatter followed by an all-gather,
    2 (n-1)/n x 236 B per Gaussian per rank), but the 28 B per element of optimizer traffic and the two moment buffers
    (2 x 236 B per Gaussian) are divided by the number of ranks, and a step of a few views per rank -- BASELINE.json config
    3 at 8 GPUs: ~4 views against a 236 MB exchange -- does not pay for a dense Adam pass on every rank on top.  Adam is
    element-wise, so sharding by flat index is exact: replicas stay bit-identical (every rank receives the same gathered
    bytes).  Per-tensor learning rates (scene/gaussian_model.py:152-165) apply to the intersection of a tensor's segment
    with the shard.

    params: the tensors in FlatGrads order; grads: their FlatGrads built with multiple_of = 4 x world_size (use
    ShardedAdam.make_buckets); lrs: one learning rate per tensor (mutable: `opt.lrs[i] = ...` follows a schedule).
    The parameter count must not change between steps; after densification `resized()` builds the optimizer of the new
    parameter set and carries the moments of the surviving Gaussians over (as the reference's optimizer surgery does)."""

    def __init__(self, params: Sequence[torch.Tensor], grads: FlatGrads, lrs: Sequence[float], betas=(0.9, 0.999),
                 eps: float = 1e-15, adam_fn: Optional[Callable] = None):
        self.world, self.rank = world_size(), get_rank()
        L = grads.flat.numel()
        if L % (4 * self.world) != 0:
            raise ValueError("ShardedAdam: build the gradient bucket with FlatGrads(params, multiple_of=4 * world_size)")
        self.grads, self.lrs = grads, [float(x) for x in lrs]
        if len(self.lrs) != len(grads.segments):
            raise ValueError("one learning rate per parameter tensor")
        self.params = FlatParams(params, grads)
        self.betas, self.eps, self.step_count = betas, float(eps), 0
        self.shard = L // self.world
        self.lo, self.hi = self.rank * self.shard, (self.rank + 1) * self.shard
        self.grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=grads.flat.device)
        self.exp_avg = torch.zeros_like(self.grad_shard)
        self.exp_avg_sq = torch.zeros_like(self.grad_shard)
        self.adam_fn = adam_fn or _adam_pieces_hip
        # (offset inside the shard, length, tensor index) of every tensor segment that meets this rank's shard
        self.pieces = []
        for t, (off, n) in enumerate(grads.segments):
            a, b = max(off, self.lo), min(off + n, self.hi)
            if a < b:
                self.pieces.append((a - self.lo, b - a, t))

    @staticmethod
    def make_buckets(params: Sequence[torch.Tensor]) -> FlatGrads:
        return FlatGrads(params, multiple_of=4 * world_size())

    # ---- the moments outlive a change of the parameter count -------------------------------------------------------
    def _full_moments(self):
        """Both moment buffers in the (old) flat layout on every rank: two all-gathers of the shards."""
        L = self.shard * self.world
        full = []
        for shard in (self.exp_avg, self.exp_avg_sq):
            if self.world > 1:
                out = torch.empty(L, dtype=torch.float32, device=shard.device)
                dist.all_gather_into_tensor(out, shard)
            else:
                out = shard
            full.append(out)
        return full

    @torch.no_grad()
    def resized(self, new_params: Sequence[torch.Tensor], keep_mask: torch.Tensor, lrs: Optional[Sequence[float]] = None):
        """After densify_and_prune changed the number of Gaussians: a new ShardedAdam over `new_params` (same tensors in the
        same order, each with rows = keep_mask.sum() + appended rows) whose moments are those of the SURVIVING rows --
        what the reference's _prune_optimizer / cat_tensors_to_optimizer do (scene/gaussian_model.py:273-326): pruned rows
        lose their moments, appended rows start at zero, everything else is kept, and the step count carries on.
        keep_mask: bool [P_old], True = the old row survives (in order) at the front of every new tensor.
        Collective: two all-gathers of the old moment shards (the flat-index shards of the old and the new layout do not
        line up), once per densification."""
        keep_mask = keep_mask.reshape(-1).bool()
        P_old = int(keep_mask.shape[0])
        old_segments = list(self.grads.segments)
        m_full, v_full = self._full_moments()
        grads = ShardedAdam.make_buckets(new_params)
        opt = ShardedAdam(new_params, grads, self.lrs if lrs is None else lrs, betas=self.betas, eps=self.eps, adam_fn=self.adam_fn)
        opt.step_count = self.step_count
        kept = int(keep_mask.sum())
        for full_old, new_shard in ((m_full, opt.exp_avg), (v_full, opt.exp_avg_sq)):
            new_full = torch.zeros(grads.flat.numel(), dtype=torch.float32, device=full_old.device)
            for (off_o, n_o), (off_n, n_n), p in zip(old_segments, grads.segments, new_params):
                if n_o % P_old != 0:
                    raise ValueError("every tensor must have one row per Gaussian")
                row = n_o // P_old
                if p.numel() % row != 0 or p.numel() // row < kept:
                    raise ValueError("new tensor does not hold the surviving rows")
                rows_old = full_old[off_o:off_o + n_o].view(P_old, row)
                new_full[off_n:off_n + kept * row].view(kept, row).copy_(rows_old[keep_mask])
            new_shard.copy_(new_full[opt.lo:opt.hi])
        return opt

    def state_dict(self):
        """This rank's shard of the optimizer state (moments of flat indices [lo, hi)) plus what is replicated."""
        return {"step": self.step_count, "lo": self.lo, "hi": self.hi, "world": self.world, "lrs": list(self.lrs),
                "betas": tuple(self.betas), "eps": self.eps, "exp_avg": self.exp_avg.detach().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().clone()}

    def load_state_dict(self, state):
        if (state["lo"], state["hi"], state["world"]) != (self.lo, self.hi, self.world):
            raise ValueError("ShardedAdam.load_state_dict: the state was saved for another shard / world size "
                             "(use full_state_dict / load_full_state_dict to move between them)")
        self.step_count, self.lrs = int(state["step"]), [float(x) for x in state["lrs"]]
        self.betas, self.eps = tuple(state["betas"]), float(state["eps"])
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])

    def full_state_dict(self):
        """The whole state on every rank (two all-gathers), independent of the world size it was trained with."""
        m, v = self._full_moments()
        return {"step": self.step_count, "lrs": list(self.lrs), "betas": tuple(self.betas), "eps": self.eps,
                "segments": list(self.grads.segments), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}

    def load_full_state_dict(self, state):
        if [n for _, n in state["segments"]] != [n for _, n in self.grads.segments]:
            raise ValueError("ShardedAdam.load_full_state_dict: tensor sizes differ")
        self.step_count, self.lrs = int(state["step"]), [float(x) for x in state["lrs"]]
        self.betas, self.eps = tuple(state["betas"]), float(state["eps"])
        for full, shard in ((state["exp_avg"], self.exp_avg), (state["exp_avg_sq"], self.exp_avg_sq)):
            mine = torch.zeros(self.shard * self.world, dtype=torch.float32, device=shard.device)
            for (off_s, n), (off_n, _) in zip(state["segments"], self.grads.segments):     # paddings may differ with the world size
                mine[off_n:off_n + n].copy_(full[off_s:off_s + n])
            shard.copy_(mine[self.lo:self.hi])

    @torch.no_grad()
    def reduce_and_update(self):
        """First half of step(): reduce-scatter of the bucket and Adam on this rank's shard (no parameter leaves the rank yet)."""
        self.step_count += 1
        flat_g, flat_p = self.grads.flat, self.params.flat
        if self.world > 1:
            dist.reduce_scatter_tensor(self.grad_shard, flat_g, op=dist.ReduceOp.SUM)
            g = self.grad_shard
        else:
            g = flat_g
        p_shard = flat_p[self.lo:self.hi]
        pieces = [(p_shard[o:o + n], g[o:o + n], self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], self.lrs[t])
                  for o, n, t in self.pieces]
        self.adam_fn(pieces, self.betas[0], self.betas[1], self.eps, self.step_count)

    @torch.no_grad()
    def gather(self, async_op: bool = False):
        """Second half: all-gather of the flat parameter buffer, in place (rank r's input IS slice r of the output).
        async_op: the collective is left in flight and its handle returned -- the caller decides what may run before the
        parameters of THIS bucket are needed again (None on one rank)."""
        if self.world > 1:
            flat_p = self.params.flat
            p_shard = flat_p[self.lo:self.hi]
            if async_op:
                return dist.all_gather_into_tensor(flat_p, p_shard, async_op=True)
            dist.all_gather_into_tensor(flat_p, p_shard)
        return None

    @torch.no_grad()
    def step(self, async_gather: bool = False):
        """Gradients of this rank's views are in self.grads.flat (accumulated); afterwards every rank holds the updated
        parameters.  The bucket is left as it was (zero it before the next step: FlatGrads.zero_).  async_gather: return the
        handle of the parameter all-gather instead of joining it (None on one rank)."""
        self.reduce_and_update()
        return self.gather(async_gather)


class SplitShardedAdam:
    """ShardedAdam over TWO buckets so that the parameter all-gather comes back in the order the next step needs it:

        geometry   means3D, scales, rotations, opacity   11 floats =  44 B per Gaussian   (19 % of the exchange)
        appearance sh                                    48 floats = 192 B per Gaussian   (81 %)

    Each bucket is sharded by flat index over all ranks and stepped like ShardedAdam (reduce-scatter, Adam on the shard,
    all-gather); Adam is element-wise, so the parameters after a step are bit for bit those of one ShardedAdam over all five
    tensors, whatever the shard boundaries (tests/test_parallel_gloo.py).  step() returns once the GEOMETRY parameters are
    complete on every rank; the appearance all-gather is still in flight (RCCL runs it on its own stream) and `wait()` joins
    it.  Everything that reads only geometry may be issued in between -- frustum culling / mark_visible of the next step's
    views (R/gaussian_renderer: `visibility_filter`), the densification statistics and their reduction, densify_and_prune's
    masks -- and the next step's first preprocess has to follow `wait()` only because the fused kernel also evaluates the SH
    colour of the survivors (preprocess.hip).  `bytes_per_step` gives what a rank puts on its links per phase (ring model)."""

    GEOMETRY = ("means3D", "scales", "rotations", "opacity")

    def __init__(self, named_params, lrs, betas=(0.9, 0.999), eps: float = 1e-15, adam_fn: Optional[Callable] = None):
        """named_params: {"means3D", "scales", "rotations", "opacity", "sh"} -> parameter tensors; lrs: {name: learning rate}."""
        order = list(self.GEOMETRY) + ["sh"]
        self.named = {k: named_params[k] for k in order}
        geo = [self.named[k] for k in self.GEOMETRY]
        self.grads_geometry = ShardedAdam.make_buckets(geo)
        self.grads_appearance = ShardedAdam.make_buckets([self.named["sh"]])
        self.geometry = ShardedAdam(geo, self.grads_geometry, [lrs[k] for k in self.GEOMETRY], betas, eps, adam_fn)
        self.appearance = ShardedAdam([self.named["sh"]], self.grads_appearance, [lrs["sh"]], betas, eps, adam_fn)
        self._pending = None

    def zero_grad(self):
        self.grads_geometry.zero_()
        self.grads_appearance.zero_()

    class _Buckets:
        """The two buckets behind the interface ChunkedViewStep uses of one FlatGrads (views in ORDER, zero_); there is no
        single flat buffer: the exchange is this optimizer's (ChunkedViewStep.run(reduce=False))."""

        def __init__(self, geometry: FlatGrads, appearance: FlatGrads):
            self.parts = (geometry, appearance)
            self.views = list(geometry.views) + list(appearance.views)
            self.segments = None
            self.flat = None

        def zero_(self):
            for b in self.parts:
                b.zero_()

    @property
    def grads(self):
        return SplitShardedAdam._Buckets(self.grads_geometry, self.grads_appearance)

    @torch.no_grad()
    def step(self):
        """Both buckets: reduce-scatter + Adam on the shard; all-gather of the geometry (joined here), all-gather of the
        appearance left in flight (joined by wait(), or by the next step()).

        ORDER OF ISSUE matters: RCCL runs the collectives of a process group in the order they were issued, on one
        communicator stream -- an all-gather issued early holds up everything issued after it, whatever `async_op` says.  So
        the geometry bucket goes first and completely (reduce-scatter, Adam, all-gather: 19 % of the bytes), then the
        appearance bucket's reduce-scatter and Adam, and its all-gather is the LAST collective issued: joining the geometry
        waits for nothing of the appearance bucket, and what is left in flight when step() returns really is the SH gather
        (rounds 4-5 issued it first; on RCCL the geometry join then waited for it and the window was empty -- ADVICE r5)."""
        self.wait()
        self.geometry.reduce_and_update()
        self.geometry.gather()
        self.appearance.reduce_and_update()
        self._pending = self.appearance.gather(async_op=True)
        return self._pending

    def wait(self):
        """Join the appearance all-gather of the last step (no-op when none is in flight)."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    @property
    def bytes_per_step(self):
        n = world_size()
        g, a = self.grads_geometry.flat.numel() * 4, self.grads_appearance.flat.numel() * 4
        half = lambda b: int((n - 1) * b // n)                 # one reduce-scatter or one all-gather of b bytes (ring)
        return {"reduce_scatter": half(g) + half(a), "all_gather_geometry_joined_in_step": half(g),
                "all_gather_appearance_left_in_flight": half(a),
                "overlap_window": "from the return of step() to wait(): the appearance all-gather (81 % of the gather half)"}


REDUCE_CHUNKS = 1


class ChunkedViewStep:
    """The multi-view step of one rank with its gradient exchange: zero the bucket, render every view forward + backward into it
    (ViewBatch: lr_views_accumulate), all-reduce the bucket.

    `chunks` > 1 splits the rank's views into consecutive groups with a bucket each and starts a group's all-reduce as soon as
    the group is enqueued (RCCL works on its own stream) while the next group renders; the buckets are summed at the end.
    It is NOT the default and cannot pay on this path: gradients cannot be reduced per parameter tensor as they become final --
    every tensor is written by every view's last kernel -- so every group reduces a FULL bucket, and the step is
    c1 + max(ar, c2) + ar against c + ar for one exchange after the last view (ar = one all-reduce, c = c1 + c2 the rendering):
    never shorter, and twice the bytes on the xGMI links in the regime where the step is communication bound (a few views per
    rank).  Kept for A/B runs on a node; with chunks = 1 (default) this is exactly ViewBatch + FlatGrads.all_reduce.

    named_params: {"means3D", "scales", "rotations", "opacity", "sh"} -> parameter tensors (their .grad become views of
    the primary bucket, `self.grads`)."""

    ORDER = ("means3D", "scales", "rotations", "opacity", "sh")

    def __init__(self, cams, grad_colors, named_params, sh_degree, bg, binning_capacity, n_streams=2, chunks=None,
                 targets=None, lambda_dssim=0.2, grads: Optional[FlatGrads] = None):
        """grads: a bucket built by the caller over the parameters in ORDER (ShardedAdam.make_buckets pads it for the
        reduce-scatter); default: a fresh FlatGrads."""
        self.named = {k: named_params[k] for k in self.ORDER}
        self.grads = grads if grads is not None else FlatGrads([self.named[k] for k in self.ORDER])
        n = len(cams)
        chunks = (REDUCE_CHUNKS if world_size() > 1 else 1) if chunks is None else chunks
        chunks = max(1, min(int(chunks), n))
        self.buckets = [self.grads] + [FlatGrads.like(self.grads) for _ in range(chunks - 1)]
        bounds = [round(i * n / chunks) for i in range(chunks + 1)]
        self.batches = []
        for i in range(chunks):
            sl = slice(bounds[i], bounds[i + 1])
            self.batches.append(ViewBatch(cams[sl], None if grad_colors is None else grad_colors[sl], sh_degree, bg,
                                          binning_capacity, n_streams=n_streams,
                                          targets=None if targets is None else targets[sl], lambda_dssim=lambda_dssim))

    def _acc(self, bucket, means2D_acc):
        d = {k: v for k, v in zip(self.ORDER, bucket.views)}
        d["means2D"] = means2D_acc
        return d

    def run(self, means2D_acc, reduce: bool = True):
        """Zeroes the buckets, renders every group forward+backward into its bucket, all-reduces (overlapped) and leaves
        the sum over all ranks and views in self.grads (the parameters' .grad).  means2D_acc [P,3] accumulates the
        screen-space gradients of THIS rank's views (densification statistics are reduced separately).
        reduce=False: no collective here -- self.grads holds THIS rank's sum and the exchange is the optimizer's
        (ShardedAdam: reduce-scatter, sharded Adam, all-gather)."""
        p = self.named
        works = []
        for b in self.buckets:
            b.zero_()
        for batch, bucket in zip(self.batches, self.buckets):
            batch.run(p["means3D"], p["opacity"], p["scales"], p["rotations"], p["sh"], self._acc(bucket, means2D_acc))
            if world_size() > 1 and reduce:
                works.append(dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        for b in self.buckets[1:]:
            self.grads.flat.add_(b.flat)

    def check(self):
        for b in self.batches:
            b.check()

    def set_streams(self, n):
        for b in self.batches:
            b.n_streams = int(n)


def all_reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor,
                                   max_radii2D: torch.Tensor):
    """SUM, SUM, MAX across ranks (in place)."""
    if world_size() == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


def densify_and_prune_synchronised(model, max_grad, min_opacity, extent, max_screen_size, seed: int):
    """densify_and_prune of a data-parallel run: the statistics every rank accumulated over ITS views are
    reduced first -- xyz_gradient_accum and denom SUM, max_radii2D MAX, what one process would have accumulated over all
    views (R/scene/gaussian_model.py:405-407, R/luciddreamer.py:310-311) -- and densify_and_split's torch.normal samples
    (:359-361) are drawn from generators every rank seeds identically for the duration of the call, so all replicas
    clone, split and prune the same Gaussians into bit-identical parameter sets.  `seed` must be the same on every rank
    (e.g. the iteration number).  Works with the reference's GaussianModel and with luciddreamer_amd.densify's methods."""
    all_reduce_densification_stats(model.xyz_gradient_accum, model.denom, model.max_radii2D)
    dev = model.get_xyz.device
    with torch.random.fork_rng(devices=[dev] if dev.type == "cuda" else []):
        torch.manual_seed(int(seed))                     # host and every device generator
        model.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)


def ring_allreduce_bytes(n_floats: int, world: Optional[int] = None) -> int:
    """Bytes every rank puts on its links for one dense fp32 all-reduce of n_floats (reduce-scatter + all-gather halves)."""
    world = world_size() if world is None else world
    return 0 if world <= 1 else int(2 * (world - 1) * n_floats * 4 // world)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None):
    """dist.all_to_all_single; over gloo with device tensors (several ranks sharing one GPU: LR_DIST_BACKEND=gloo, debugging
    only) the exchange is staged through the host -- gloo's all-to-all takes host tensors only."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


@torch.no_grad()
def sparse_rows_all_reduce(views: Sequence[torch.Tensor], touched: Optional[torch.Tensor] = None) -> dict:
    """All-reduce(SUM) of per-Gaussian gradient tensors whose LOCAL contribution is sparse in the rows.

    A rank's views of a step touch a small part of the scene (C3: 4 views of a rotate360 path see ~9 % of the Gaussians
    each), so the gradient it has accumulated is zero in most rows, while the SUM over all ranks is dense.  The dense
    all-reduce is a reduce-scatter followed by an all-gather; here the reduce-scatter half is replaced:

      rows are owned in contiguous blocks (rank r: rows [r P / n, (r + 1) P / n) of EVERY tensor);
      every rank sends each row it touched -- index + all its floats, 4 + 236 B at SH degree 3 -- to the row's owner
      (one all-to-all of indices, one of payloads; the counts go first);
      the owner adds what it received to its own rows, source by source in rank order (indices are unique within a source:
      no atomics, bit-repeatable);
      the owners' blocks are all-gathered (in place when n divides P, else one broadcast per owner).

    Every rank ends with the owners' bytes, so replicas stay bit-identical; the sum differs from the ring all-reduce's only
    in association.  views: tensors [P, ...] (e.g. FlatGrads.views), modified in place.  touched: bool [P], rows with a
    non-zero local contribution (default: computed here, one pass over the tensors).  Returns the bytes this rank sent."""
    n, r = world_size(), get_rank()
    P = int(views[0].shape[0])
    rows = [v.view(P, -1) for v in views]
    K = sum(x.shape[1] for x in rows)
    if n == 1:
        return {"sent_rows": 0, "bytes_sent": 0, "dense_equivalent_bytes": 0}
    dev = rows[0].device
    if touched is None:
        touched = torch.zeros(P, dtype=torch.bool, device=dev)
        for x in rows:
            touched |= (x != 0).any(dim=1)
    bounds = [i * P // n for i in range(n + 1)]
    idx = touched.nonzero().view(-1)                                      # ascending
    cuts = torch.searchsorted(idx, torch.tensor(bounds, device=dev, dtype=idx.dtype)).tolist()     # the step's one host read
    send_counts = [0 if d == r else cuts[d + 1] - cuts[d] for d in range(n)]
    sel = torch.cat([idx[cuts[d]:cuts[d + 1]] for d in range(n) if d != r]) if n > 1 else idx[:0]
    payload = torch.cat([x[sel] for x in rows], dim=1) if sel.numel() else torch.zeros((0, K), dtype=torch.float32, device=dev)
    cnt_in = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    cnt_out = torch.zeros(n, dtype=torch.int64, device=dev)
    _all_to_all(cnt_out, cnt_in)
    recv_counts = cnt_out.tolist()
    n_recv = sum(recv_counts)
    idx_recv = torch.zeros(n_recv, dtype=torch.int32, device=dev)
    _all_to_all(idx_recv, sel.to(torch.int32), recv_counts, send_counts)
    pay_recv = torch.zeros((n_recv, K), dtype=torch.float32, device=dev)
    _all_to_all(pay_recv, payload.contiguous(), recv_counts, send_counts)
    off = 0
    for src in range(n):                                                  # fixed order over the sources
        c = recv_counts[src]
        if c:
            ii = idx_recv[off:off + c].long()
            col = 0
            for x in rows:
                w = x.shape[1]
                x.index_add_(0, ii, pay_recv[off:off + c, col:col + w])
                col += w
            off += c
    lo, hi = bounds[r], bounds[r + 1]
    if P % n == 0:
        for x in rows:
            dist.all_gather_into_tensor(x.view(-1), x[lo:hi].reshape(-1))   # in place: this rank's block IS slice r of the output
    else:
        for x in rows:
            for o in range(n):
                dist.broadcast(x[bounds[o]:bounds[o + 1]], src=o)
    # bytes THIS rank puts on its links: its touched rows to their owners, then (ring all-gather) n - 1 blocks of P K / n floats
    sent_rows = int(sel.numel())
    a2a = sent_rows * (K * 4 + 4)
    gather = (n - 1) * ((P + n - 1) // n) * K * 4
    return {"sent_rows": sent_rows, "bytes_all_to_all": a2a, "bytes_all_gather": gather, "bytes_sent": a2a + gather,
            "dense_equivalent_bytes": ring_allreduce_bytes(P * K, n)}


def _calling_thread_backward():
    """Context: backward passes started inside run their nodes on THIS thread (torch.autograd.set_multithreading_enabled(False))
    instead of being handed to the autograd engine's device thread -- every node of this path only enqueues kernels, and the
    hand-off (plus the two threads taking turns at the interpreter lock) was 110-130 us of a view's 180 us of host time."""
    import contextlib
    if hasattr(torch.autograd, "set_multithreading_enabled"):
        return torch.autograd.set_multithreading_enabled(False)
    return contextlib.nullcontext()


def _direct_backward(out: torch.Tensor, grad_output: torch.Tensor) -> bool:
    """Backward of ONE rasterizer node without the autograd engine: `out` must be the image returned by the rasterizer op,
    every differentiable input of the node a LEAF (its next function an AccumulateGrad, or none), and fused gradient
    accumulation on, so that the node's kernels add into the leaves' .grad in place.  The node is called on the calling thread
    and current stream (= the stream of its forward); a gradient it returns for a leaf it could not accumulate into (no .grad
    yet, unsuitable layout) is added the way AccumulateGrad would.  Returns False -- nothing done -- when the shape of the
    graph is anything else: the caller goes through the engine then.  Tensor hooks on the leaves are not run (as with fused
    accumulation in general, config.set_fused_grad_accumulation)."""
    from . import config
    fn = out.grad_fn
    if fn is None or not config.fused_grad_accumulation() or "Rasterize" not in type(fn).__name__ + fn.name():
        return False
    # only the COMPILED node (csrc/torch_ext.cpp RasterizeFn, a torch::autograd::CppNode) can be called like a function;
    # the nodes of Python autograd.Functions (the raw-parameter path render_raw, the debug-mode operator) are not callable
    # objects on torch 2.10 ('...Backward' object is not callable): those views go through the engine
    if "RasterizeFn" not in fn.name() or not callable(fn):
        return False
    leaves = []
    for nxt, _ in fn.next_functions:
        if nxt is None:
            leaves.append(None)
        elif type(nxt).__name__ == "AccumulateGrad":
            leaves.append(nxt.variable)
        else:
            return False
    grads = fn(grad_output, None, None, None)                    # the compiled node's outputs: (color, radii, depth, geom)
    if not isinstance(grads, (tuple, list)):
        grads = (grads,)
    with torch.no_grad():
        for leaf, g in zip(leaves, grads):
            if leaf is not None and g is not None:
                leaf.grad = g if leaf.grad is None else leaf.grad.add_(g)
    return True


class ViewStreams:
    """Round-robin HIP streams for the consecutive views of one rank.

    The forward of a view is a chain of ~25 short, launch-latency-bound kernels (sorts, scans) followed by
    the blend; the backward is two long kernels.  Putting consecutive views on alternating streams lets the
    hardware run forward(i+1) underneath backward(i).  Backward passes are chained with an event because they
    accumulate into the same gradient buffers (FlatGrads); forwards only read the parameters.
    """

    def __init__(self, device, n_streams: int = 2, group: int = 6, direct: bool = True, on_overflow: str = "recover"):
        """group: views that share ONE pass of the autograd engine when run_view is given `grad_output` instead of a
        backward function; direct: call the rasterizer's backward node on the calling thread instead when that is
        equivalent (see run_view); on_overflow: the policy of the step's views -- "recover" (default: no view is lost, end_step
        waits for the last view's header) or "drop" / "raise" (config.py: never waits)."""
        self.device = device
        self.group = max(1, int(group))
        self.direct = bool(direct)
        self.on_overflow = on_overflow
        self._deferred = []
        self.streams = [torch.cuda.Stream(device) for _ in range(max(1, n_streams))]
        # "backward of view i done" events, re-used round robin: an event is waited on (by the next view) right after it
        # is recorded, so a ring of two would do; one per stream keeps it obvious
        self._events = [torch.cuda.Event() for _ in range(len(self.streams) + 1)]
        self._i = 0
        self._prev_bwd = None
        self._caller = None
        self.waited_s = 0.0              # total time end_step() spent waiting for the last view's header ("recover")

    def begin_step(self):
        from . import _lib, config
        if getattr(self, "_policy", None) is not None:       # a step that never reached end_step() (exception in user code)
            self._pop_policy()
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)
        self._prev_bwd = None
        self._caller = cur
        # several views in flight: a forward must not wait for its own header copy (config "verify" does).  "recover": every
        # view's header is examined at end_step() and a view that overflowed its binning buffer -- the device-side guard
        # zeroed its gradients -- is run again in exact mode there, so no view of the step is lost
        self._policy = config.overflow_policy(self.on_overflow, _owner=self)
        self._policy.__enter__()
        self._views = []
        self._deferred = []
        self.recovered = 0
        # kernel shapes for a GPU shared by several views (render_bwd.hip blend_shape); never a correctness input
        _lib.tune_set("views_in_flight", len(self.streams))
        # the step's accumulate-mode backward passes share the library's interleaved accumulator of the five small rows
        # (lr_step_begin: what lr_views_accumulate does internally); end_step hands the rows to the .grad tensors
        L = _lib.lib()
        with torch.cuda.device(self.device):
            if L.lr_step_begin() < 0:                        # a step that never reached end_step(): its rows are dropped, not
                L.lr_step_abort()                            # flushed (its target tensors may be gone: ADVICE r4)
                L.lr_step_begin()
        self._step_open = True

    def _close_step(self, stream, abort=False):
        from . import _lib
        if getattr(self, "_step_open", False):
            self._step_open = False
            with torch.cuda.device(self.device):
                if abort:
                    _lib.lib().lr_step_abort()
                elif _lib.lib().lr_step_end(stream.cuda_stream) < 0:
                    _lib.raise_for(-1, "lr_step_end")

    def _pop_policy(self):
        """Leaves the step's policy / tuning state; a step still open here did not reach end_step() (it closes the step itself
        first): an exception in the caller's loop.  ABORTED STEP = UNDEFINED GRADIENTS: the five small rows of the views already
        run (means2D, opacity, means3D, scales, rotations) were still in the library's interleaved accumulator and are dropped,
        while their SH / colour gradients went straight into .grad and stay there.  A caller that catches the exception must zero
        the gradient tensors before the next step (FlatGrads.zero_()) and must not step the optimizer on what is left."""
        from . import _lib
        self._close_step(torch.cuda.current_stream(self.device), abort=True)
        if getattr(self, "_policy", None) is not None:
            self._policy.__exit__(None, None, None)
            self._policy = None
            _lib.tune_set("views_in_flight", -1)

    def run_view(self, forward_fn: Callable, backward_fn: Optional[Callable] = None, grad_output: Optional[torch.Tensor] = None):
        """forward_fn() -> the view's output tensor, issued on the next stream of the ring.  Then EITHER
        backward_fn(out): the view's backward right away (its own pass of the autograd engine), chained behind the previous
            view's by an event; OR
        grad_output: dL/d out -- the backward is DEFERRED and shares one engine pass (`torch.autograd.backward` over the
            group's outputs) with up to `group` consecutive views.  The engine hands a backward to its device thread and waits
            for it: ~130 us of host time per call on this path (profiles/r04d_host_breakdown.txt), more than a view's forward
            and backward launches together; one pass per group leaves ~20 us per view.  Every node still runs on the stream
            of its own forward, and accumulating backward passes are chained on the device by the binding itself
            (csrc/torch_ext.cpp AccumulateChain), so the gradients are those of the per-view form.  With `direct` (default)
            and `out` the rasterizer's own output over LEAF inputs under fused gradient accumulation, there is nothing for the
            engine to do at all -- one node, gradients added in place by its kernels -- and the node is called right here, on
            this thread (_direct_backward): ~25 us instead of the engine's ~110 us per view, no grouping needed."""
        from . import config
        if (backward_fn is None) == (grad_output is None):
            raise ValueError("give exactly one of backward_fn / grad_output")
        s = self.streams[self._i % len(self.streams)]
        # set_stream instead of the `with torch.cuda.stream(s)` context: the context manager's save / restore per view is
        # ~10 us of host time on a path that is host bound; end_step() puts the caller's stream back
        torch.cuda.set_stream(s)
        try:
            config.take_last_entry()
            # grad_output known up front: the rasterizer call inside forward_fn may run its backward right behind its forward
            # (one call into the binding, no autograd node -- rasterizer.rasterize_gaussians); if it did, nothing is left to do
            config.offer_grad_output(grad_output if (grad_output is not None and self.direct) else None)
            try:
                out = forward_fn()
            finally:
                taken = config.grad_output_taken()
            entry = config.take_last_entry()                 # this view's header entry, if its forward was an async one
            if taken:
                redo = lambda o, g=grad_output: torch.autograd.backward([o], [g])     # (end_step: a lost view, run again)
            elif backward_fn is not None:
                if self._prev_bwd is not None:
                    s.wait_event(self._prev_bwd)
                with _calling_thread_backward():
                    backward_fn(out)
                ev = self._events[self._i % len(self._events)]
                ev.record(s)
                self._prev_bwd = ev
                redo = backward_fn
            else:
                if not (self.direct and _direct_backward(out, grad_output)):
                    self._deferred.append((out, grad_output))
                    if len(self._deferred) >= self.group:
                        self._flush()
                redo = lambda o, g=grad_output: torch.autograd.backward([o], [g])     # (end_step: a lost view, run again)
            if entry is not None:
                self._views.append((entry, forward_fn, redo))
        except BaseException:
            if self._caller is not None:
                torch.cuda.set_stream(self._caller)
            self._deferred = []
            self._pop_policy()
            raise
        self._i += 1

    def _flush(self):
        if self._deferred:
            outs, grads = zip(*self._deferred)
            self._deferred = []
            with _calling_thread_backward():
                torch.autograd.backward(list(outs), list(grads))

    def end_step(self):
        from . import config
        try:
            self._flush()
        except BaseException:
            if self._caller is not None:
                torch.cuda.set_stream(self._caller)
            self._pop_policy()
            raise
        cur = self._caller if self._caller is not None else torch.cuda.current_stream(self.device)
        torch.cuda.set_stream(cur)
        for s in self.streams:
            cur.wait_stream(s)
        self._close_step(cur)                                # after every view of the step, before any re-run
        self._caller = None
        views, self._views = getattr(self, "_views", []), []
        try:
            if views:
                # The header copies complete when the LAST view's compaction scan has run -- its blend and backward are still
                # queued behind, so the GPU stays busy while the host looks.  A view whose instance count exceeded its
                # buffer contributed zero gradients; it is run again here, exact mode, on the caller's stream.
                import time
                t0 = time.perf_counter()
                config.drain()
                self.waited_s += time.perf_counter() - t0     # host time spent BLOCKED on the last header (not host work)
                lost = [v for v in views if v[0][3]]
                if lost:
                    with config.force_exact(), config.overflow_policy("verify"):
                        for _, fwd, bwd in lost:
                            bwd(fwd())
                    self.recovered = len(lost)
                    config.recovered_views += len(lost)
        finally:
            self._pop_policy()
        return self.recovered


class ViewBatch:
    """One C call per step for a fixed list of views (lr_views_accumulate): forward + backward of every view,
    gradients accumulated in place, views alternated over internal HIP streams.  The per-view upstream
    gradients dL/dcolor are given up front (a training loop that needs the rendered image to form its loss uses
    the autograd op, optionally with ViewStreams, instead).

    cams: objects with world_view_transform, full_proj_transform, camera_center (device tensors), FoVx, FoVy,
          image_width, image_height (e.g. cameras.MiniCam); all views share one resolution.
    """

    def __init__(self, cams: Sequence, grad_colors: Optional[Sequence[torch.Tensor]], sh_degree: int, bg: torch.Tensor,
                 binning_capacity: int, n_streams: int = 2, scale_modifier: float = 1.0,
                 targets: Optional[Sequence[torch.Tensor]] = None, lambda_dssim: float = 0.2):
        """grad_colors: fixed upstream gradients dL/dcolor per view, OR targets: ground-truth images per view, in which
        case every view's L1 + DSSIM loss against its target is formed inside the call (lr_views_train_accumulate) and
        `self.losses` ([n,3] device tensor: loss, l1, ssim per view) is filled by run()."""
        import ctypes
        import math
        from . import _lib
        self._lib = _lib
        self.L = _lib.lib()
        self.cams = list(cams)
        self.n = len(self.cams)
        if (grad_colors is None) == (targets is None):
            raise ValueError("give exactly one of grad_colors / targets")
        assert self.n == len(grad_colors if targets is None else targets) and self.n > 0
        self.W, self.H = int(self.cams[0].image_width), int(self.cams[0].image_height)
        self.device = self.cams[0].world_view_transform.device
        self.degree, self.scale_modifier = int(sh_degree), float(scale_modifier)
        self.capacity, self.n_streams = int(binning_capacity), int(n_streams)
        self.bg = bg.to(self.device).contiguous()
        keep = []

        def ptr_array(tensors):
            ts = [t.to(self.device).contiguous() for t in tensors]
            keep.extend(ts)
            return (ctypes.c_void_p * self.n)(*[t.data_ptr() for t in ts])
        self._views = ptr_array([c.world_view_transform for c in self.cams])
        self._projs = ptr_array([c.full_proj_transform for c in self.cams])
        self._campos = ptr_array([c.camera_center for c in self.cams])
        self.train = targets is not None
        self.lambda_dssim = float(lambda_dssim)
        self._grads = ptr_array(grad_colors if not self.train else targets)     # per-view dL/dcolor, or target images
        self.losses = torch.zeros((self.n, 3), dtype=torch.float32, device=self.device) if self.train else None
        self._tanx = (ctypes.c_float * self.n)(*[math.tan(c.FoVx * 0.5) for c in self.cams])
        self._tany = (ctypes.c_float * self.n)(*[math.tan(c.FoVy * 0.5) for c in self.cams])
        self._keep = keep
        self._ws = None
        self._ws_key = None

    def run(self, means3D, opacities, scales, rotations, shs, acc: dict):
        """acc: {"means3D", "means2D", "opacity", "sh", "scales", "rotations"} -> contiguous float32 tensors that are
        accumulated into (e.g. the .grad views of a FlatGrads bucket)."""
        P, M = int(means3D.shape[0]), int(shs.shape[1])
        key = (P, self.capacity, self.n_streams)      # the workspace is sized per stream slot: set_streams() re-allocates
        if self._ws_key != key:
            size_fn = self.L.lr_views_train_workspace_bytes if self.train else self.L.lr_views_workspace_bytes
            nbytes = size_fn(P, self.W, self.H, self.capacity, self.n_streams)
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
            self._ws_key = key
        for t in (means3D, opacities, scales, rotations, shs, *acc.values()):
            if not (t.is_cuda and t.dtype is torch.float32 and t.is_contiguous()):
                raise RuntimeError("ViewBatch.run needs contiguous float32 tensors on the HIP device")
        with torch.cuda.device(self.device):
            self._run(P, M, means3D, opacities, scales, rotations, shs, acc)

    def _run(self, P, M, means3D, opacities, scales, rotations, shs, acc):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.train:
            rc = self.L.lr_views_train_accumulate(
                self.n, self._views, self._projs, self._campos, self._tanx, self._tany, P, self.degree, M,
                self.bg.data_ptr(), self.W, self.H, means3D.data_ptr(), shs.data_ptr(), opacities.data_ptr(),
                scales.data_ptr(), self.scale_modifier, rotations.data_ptr(), self._grads, self.lambda_dssim,
                self.losses.data_ptr(), None, None, acc["means2D"].data_ptr(), acc["opacity"].data_ptr(),
                acc["means3D"].data_ptr(), acc["sh"].data_ptr(), acc["scales"].data_ptr(), acc["rotations"].data_ptr(),
                self._ws.data_ptr(), self._ws.numel(), self.capacity, self.n_streams, stream)
            if rc < 0:
                self._lib.raise_for(rc, "lr_views_train_accumulate")
            return
        rc = self.L.lr_views_accumulate(
            self.n, self._views, self._projs, self._campos, self._tanx, self._tany, P, self.degree, M,
            self.bg.data_ptr(), self.W, self.H, means3D.data_ptr(), shs.data_ptr(), None, opacities.data_ptr(),
            scales.data_ptr(), self.scale_modifier, rotations.data_ptr(), None, self._grads, None, None,
            acc["means2D"].data_ptr(), acc["opacity"].data_ptr(), None, acc["means3D"].data_ptr(), None,
            acc["sh"].data_ptr(), acc["scales"].data_ptr(), acc["rotations"].data_ptr(),
            self._ws.data_ptr(), self._ws.numel(), self.capacity, self.n_streams, stream)
        if rc < 0:
            self._lib.raise_for(rc, "lr_views_accumulate")

    def check(self):
        """Synchronise and raise if any view of the previous run() overflowed the binning capacity."""
        if self._ws is None:
            return
        P = self._ws_key[0]
        check = self.L.lr_views_train_check if self.train else self.L.lr_views_check
        with torch.cuda.device(self.device):
            rc = check(self._ws.data_ptr(), P, self.W, self.H, self.capacity, self.n_streams,
                       torch.cuda.current_stream(self.device).cuda_stream)
        if rc < 0:
            self._lib.raise_for(rc, "lr_views_check")


def dp_step(views: Sequence, params: Sequence[torch.Tensor], loss_fn: Callable, grads: Optional[FlatGrads] = None,
            rank: Optional[int] = None, world: Optional[int] = None, reduce: bool = True) -> FlatGrads:
    """One data-parallel gradient step over `views`.

    loss_fn(view, view_index) -> scalar loss of that view (it renders through the rasterizer).
    Every rank calls this with the SAME `views` list; rank r processes views r, r+n, ...; after the
    single all-reduce every rank holds sum_over_all_views dLoss/dparams in grads.flat.
    """
    if grads is None:
        grads = FlatGrads(params)
    grads.zero_()
    for i in shard_views(len(views), rank, world):
        loss = loss_fn(views[i], i)
        loss.backward()
    if reduce:
        grads.all_reduce()
    return grads
