"""Run-time switches of the rasterizer that have no slot in the reference's 12-field settings tuple.

Exact mode reproduces the reference's host behaviour: one small D2H read of num_rendered per forward
(RAST/cuda_rasterizer/rasterizer_impl.cu:281-282) to size the binning buffer exactly.  It is what the first forwards
of every problem size run (`warm_calls`), what forwards that will not be differentiated run (video rendering: nothing
could repair a frame afterwards), and what `set_async(False)` / LUCID_RASTER_EXACT=1 select for everything.

Async mode (the default for forwards that will be differentiated, once a problem size has been seen `warm_calls`
times) removes that host round trip: the binning buffer is sized from the high-water mark of the instance counts seen
so far for the same (device, P, H, W) times `headroom`; every kernel takes its counts from the device-side header, and a
view that needs more instances than the buffer holds is flagged there (`overflow`), never written out of bounds.  A
non-blocking 48-byte copy of the header follows every async forward (lr_header_post: pinned memory and an event owned
by the library).  What happens to a view that overflowed is the `on_overflow` policy -- in NO case do gradients of a
truncated instance list reach the caller: the backward kernels of such a view write nothing (device-side test of the
same flag), so the worst case is a view that contributes zero, never a wrong gradient.

  "rerender" (default)  The autograd backward enqueues its (self-skipping) kernels and then looks at ITS forward's header
                        copy.  If the view overflowed it is rendered again in exact mode and THAT render is
                        differentiated: the caller gets exact-mode gradients.  (The image it already received was
                        incomplete; the mark is raised.)  With `wait=False` (default) the look is a poll: in a loop whose
                        host is the bottleneck -- the reference's training loop, 4 of its 5 ms per iteration are Python
                        and launch time -- the forward has long finished when backward() runs and the header is there;
                        if the host runs AHEAD of the GPU and the copy has not arrived, the ticket goes to the deferred
                        check below (that view then counts as "drop").  `wait=True` blocks on the copy instead: every
                        overflowed view is re-rendered, at the price of a host stall per backward (measured on the
                        unchanged reference loop at 1 M Gaussians, 512 x 512: 4.98 -> 5.75 ms per iteration).
  "drop"                Never waits -- for callers that keep several views in flight (parallel.ViewStreams selects it
                        for its own duration).  An overflowed view's gradients are zero (device-side guard); the
                        deferred check warns and raises the mark.
  "raise"               Like "drop", but the deferred check raises RuntimeError on a later call or at drain().
"""
import os
import warnings
import weakref


_async = os.environ.get("LUCID_RASTER_EXACT", "0") != "1"
_fused_accumulate = False
_headroom = 1.3
_hwm = {}            # (device, P, H, W) -> largest instance count observed
_pending = []        # [[ticket of lr_header_post, key, owner]]; owner: weakref to the autograd ctx token that will claim it
_CHECK_EVERY = 1      # "drop" / "raise": every k-th async forward gets its header copied back and checked
_calls = 0
_warm_calls = 2
_seen = {}           # key -> forwards seen
_on_overflow = "rerender"
_wait = False        # "rerender": block on the forward's header copy in backward (True) or poll it once (False)
_override = []       # stack of temporary policies (parallel.ViewStreams)
dropped_views = 0    # views whose gradients were zeroed by the device-side guard (policies "drop" / "raise")
rerendered_views = 0

POLICIES = ("rerender", "drop", "raise")


def set_async(enabled: bool, headroom: float = 1.3, check_every: int = 1, warm_calls: int = 2,
              on_overflow: str = "rerender", wait: bool = False):
    """enabled: False = exact mode for every forward.
    headroom: binning capacity = high-water mark x headroom (+ 4096).
    warm_calls: the first this-many forwards of a (device, P, H, W) run in exact mode and feed the high-water mark.
    on_overflow: "rerender" (default), "drop" or "raise" -- see the module docstring; wait: "rerender" blocks in backward
    until the forward's header is on the host (strict: every overflowed view is re-rendered) instead of polling once.
    check_every ("drop" / "raise" only): every k-th async forward has its header copied back and examined on a later
    call; k > 1 samples (views in between can overflow unnoticed -- their gradients are still zero, never wrong -- and do
    not feed the mark)."""
    global _async, _headroom, _CHECK_EVERY, _warm_calls, _on_overflow, _wait
    if on_overflow not in POLICIES:
        raise ValueError(f"on_overflow must be one of {POLICIES}")
    _async = bool(enabled)
    _headroom = float(headroom)
    _CHECK_EVERY = max(1, int(check_every))
    _warm_calls = max(1, int(warm_calls))
    _on_overflow = on_overflow
    _wait = bool(wait)
    if not enabled:
        drain()


class overflow_policy:
    """Context manager: a temporary policy (parallel.ViewStreams keeps views in flight and must not wait in backward)."""

    def __init__(self, policy):
        if policy not in POLICIES:
            raise ValueError(f"policy must be one of {POLICIES}")
        self.policy = policy

    def __enter__(self):
        _override.append(self.policy)
        return self

    def __exit__(self, *exc):
        _override.pop()
        return False


def current_policy() -> str:
    return _override[-1] if _override else _on_overflow


def set_fused_grad_accumulation(enabled: bool):
    """When on, the backward of the rasterizer op adds the gradient of every LEAF input whose .grad tensor
    already exists (contiguous float32) directly into that .grad inside the HIP kernel -- touching only the
    rows of visible Gaussians -- and returns None for it, instead of materialising a dense gradient that
    autograd then adds in a separate pass.  Numerically this is autograd's own `grad += new` (same sum per
    element); tensor hooks registered on those leaves are NOT run, hence opt-in.  Used by the data-parallel
    multi-view step (luciddreamer_amd.parallel / bench.py)."""
    global _fused_accumulate
    _fused_accumulate = bool(enabled)


def fused_grad_accumulation() -> bool:
    return _fused_accumulate


def is_async() -> bool:
    return _async


def reset():
    drain()
    _hwm.clear()
    _seen.clear()


def _key(means3D, rs):
    return (means3D.device.index, int(means3D.shape[0]), int(rs.image_height), int(rs.image_width))


def _digest(words, key, policy):
    """Feed the mark from a header; returns True if that view overflowed.  words[6] = instances actually needed."""
    global dropped_views
    num_instances, overflow, trap = words[6], words[1], words[2]
    if num_instances > _hwm.get(key, 0):
        _hwm[key] = num_instances
    if trap:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    if overflow and policy != "rerender":
        dropped_views += 1
        msg = (f"luciddreamer_amd async mode: a view needed {num_instances} tile instances, more than its binning "
               "capacity; its image was incomplete and its gradients were ZERO (the backward kernels skip an "
               "overflowed view). The capacity has been raised")
        if policy == "raise":
            raise RuntimeError(msg + "; re-run the view, or use on_overflow='rerender' / exact mode "
                                     "(luciddreamer_amd.config.set_async).")
        warnings.warn(msg + ".")
    return bool(overflow)


def _poll(block=False):
    """Examine completed header copies in order.  Entries that an autograd backward is going to claim ("rerender")
    are left alone while their owner is alive."""
    from . import _C
    i = 0
    while i < len(_pending):
        ticket, key, owner, policy = _pending[i]
        if owner is not None and owner() is not None:
            i += 1                               # its backward will claim it
            continue
        words = _C.header_poll(ticket, block)
        if words is None:
            break                                # in flight: everything behind it is younger
        _pending.pop(i)
        _digest(words, key, "drop" if policy == "rerender" else policy)


def drain():
    """Wait for all outstanding async forwards and surface a deferred overflow, if any."""
    _poll(block=True)


def capacity_for(means3D, rs, differentiable=True) -> int:
    """0 = exact mode for this call; otherwise the number of tile instances to size the binning buffer for.
    differentiable=False (no backward will follow): exact, because nothing could repair the image afterwards."""
    if not _async or means3D.shape[0] == 0 or not differentiable:
        return 0
    _poll()
    key = _key(means3D, rs)
    est = _hwm.get(key)
    n = _seen.get(key, 0)
    _seen[key] = n + 1
    if est is None or (_warm_calls > 1 and n < _warm_calls):
        return 0                      # first sighting(s) of this problem size: measure exactly
    return int(est * _headroom) + 4096


class _Owner:
    """Token held by an autograd ctx: while it is alive the ctx's header ticket is reserved for its backward."""
    __slots__ = ("__weakref__",)


def note_forward(means3D, rs, num_rendered, geom, capacity):
    """After a forward.  Exact forwards feed the mark directly; async ones post the header copy.  Returns None or, under
    the "rerender" policy, (ticket, owner): the caller keeps `owner` alive on its ctx and passes `ticket` to claim()."""
    if not _async or means3D.shape[0] == 0:
        return None
    key = _key(means3D, rs)
    if capacity == 0:
        if num_rendered > _hwm.get(key, 0):
            _hwm[key] = num_rendered
        return None
    policy = current_policy()
    global _calls
    if policy != "rerender":
        _calls += 1
        if _calls % _CHECK_EVERY:
            return None
    from . import _C
    # a 48-byte copy into pinned memory + an event, both owned by the library (lr_header_post): a few microseconds of host
    # time per view
    ticket = _C.header_post(geom)
    if policy != "rerender":
        _pending.append([ticket, key, None, policy])
        return None
    owner = _Owner()
    _pending.append([ticket, key, weakref.ref(owner), policy])
    return ticket, owner


def claim(ticket) -> bool:
    """From the backward of a "rerender" forward, after its kernels were enqueued: look at that forward's header, feed the
    mark, return True if the view overflowed (the caller renders it again in exact mode).  Blocks only with wait=True;
    otherwise a copy that has not arrived hands the ticket to the deferred check."""
    global rerendered_views
    from . import _C
    for i, e in enumerate(_pending):
        if e[0] == ticket:
            words = _C.header_poll(ticket, _wait)
            if words is None:
                e[2], e[3] = None, "drop"             # not there yet: the deferred check will look at it
                return False
            _pending.pop(i)
            over = _digest(words, e[1], "rerender")
            if over:
                rerendered_views += 1
            return over
    return False                                  # already examined (e.g. drain() after the graph was released)
