"""Run-time switches of the rasterizer that have no slot in the reference's 12-field settings tuple.

Exact mode (default) reproduces the reference's host behaviour: one 4-byte D2H read of num_rendered
per forward (RAST/cuda_rasterizer/rasterizer_impl.cu:281-282) to size the binning buffer exactly.

Async mode removes that host round trip so the CPU can enqueue many views ahead of the GPU:
the binning buffer is sized from the high-water mark of num_rendered seen so far for the same
(P, H, W) times a headroom factor.  Every async forward leaves its true count and an overflow flag in
the geom-buffer header; a non-blocking 32-byte copy of the header is taken after every forward (check_every=1, the
default; a larger value samples every k-th view only and can miss an overflow of the views in between) and polled on
later calls.  If a checked view overflowed, a later rasterizer call (or config.drain()) raises -- that earlier image
was incomplete -- so callers that cannot accept a deferred error keep exact mode.  The first `warm_calls` forwards of every (P, H, W) still run exact, so
that the mark is taken over several views of a camera path rather than the first one only; a training loop whose
counts keep growing (scales change, densification) can pass on_overflow="warn" to keep going with a raised
capacity instead of an exception.
"""
import warnings


_async = False
_fused_accumulate = False
_headroom = 1.3
_hwm = {}            # (device, P, H, W) -> largest num_rendered observed
_pending = []        # [(ticket of lr_header_post, key)]
_CHECK_EVERY = 1      # async mode: every k-th forward gets its header copied back and checked
_calls = 0
_warm_calls = 1
_seen = {}           # key -> forwards seen
_on_overflow = "raise"


def set_async(enabled: bool, headroom: float = 1.3, check_every: int = 1, warm_calls: int = 1,
              on_overflow: str = "raise"):
    """check_every: every k-th async forward has its header (true instance count, overflow flag) copied back
    without blocking and examined on a later call; 1 (default) checks every view, k > 1 only every k-th (views in
    between can overflow unnoticed and do not feed the high-water mark).
    warm_calls: the first this-many forwards of a (P, H, W) run in exact mode and feed the high-water mark.
    on_overflow: "raise" (default) or "warn" when a deferred overflow is discovered."""
    global _async, _headroom, _CHECK_EVERY, _warm_calls, _on_overflow
    if on_overflow not in ("raise", "warn"):
        raise ValueError("on_overflow must be 'raise' or 'warn'")
    _async = bool(enabled)
    _headroom = float(headroom)
    _CHECK_EVERY = max(1, int(check_every))
    _warm_calls = max(1, int(warm_calls))
    _on_overflow = on_overflow
    if not enabled:
        drain()


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


def _poll(block=False):
    from . import _C
    while _pending:
        ticket, key = _pending[0]
        words = _C.header_poll(ticket, block)
        if words is None:
            break
        _pending.pop(0)
        num_rendered, overflow, trap = words[6], words[1], words[2]                 # [6] = instances actually emitted
        if num_rendered > _hwm.get(key, 0):
            _hwm[key] = num_rendered
        if trap:
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
        if overflow and _on_overflow == "warn":
            warnings.warn(f"luciddreamer_amd async mode: a view needed {num_rendered} tile instances, more than its "
                          "binning capacity; its image/gradient was incomplete (capacity has been raised)")
        elif overflow:
            raise RuntimeError(
                f"luciddreamer_amd async mode: an earlier view needed {num_rendered} tile instances, more than its "
                "binning capacity; that image/gradient was incomplete. Re-run it (capacity has been raised) or use "
                "exact mode (luciddreamer_amd.config.set_async(False)).")


def drain():
    """Wait for all outstanding async forwards and surface a deferred overflow, if any."""
    _poll(block=True)


def capacity_for(means3D, rs) -> int:
    """0 = exact mode for this call; otherwise the number of tile instances to size the binning buffer for."""
    if not _async or means3D.shape[0] == 0:
        return 0
    _poll()
    key = _key(means3D, rs)
    est = _hwm.get(key)
    n = _seen.get(key, 0)
    _seen[key] = n + 1
    if est is None or (_warm_calls > 1 and n < _warm_calls):
        return 0                      # first sighting(s) of this problem size: measure exactly
    return int(est * _headroom) + 4096


def note_forward(means3D, rs, num_rendered, geom, capacity):
    if not _async or means3D.shape[0] == 0:
        return
    key = _key(means3D, rs)
    if capacity == 0:
        if num_rendered > _hwm.get(key, 0):
            _hwm[key] = num_rendered
        return
    global _calls
    _calls += 1
    if _calls % _CHECK_EVERY:
        return
    from . import _C
    # a 32-byte copy into pinned memory + an event, both owned by the library (lr_header_post): a few microseconds of host
    # time per view instead of the ~25 the same thing cost through torch tensors, events and stream objects
    _pending.append((_C.header_post(geom), key))
