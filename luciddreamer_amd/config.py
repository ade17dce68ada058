"""Run-time switches of the rasterizer that have no slot in the reference's 12-field settings tuple.

Exact mode reproduces the reference's host behaviour: one small D2H read of num_rendered per forward
(RAST/cuda_rasterizer/rasterizer_impl.cu:281-282) in the MIDDLE of the forward, to size the binning buffer exactly; the GPU
idles from the end of the scan until the host has woken up and enqueued the remaining kernels.  It is what the first forwards
of every problem size run (`warm_calls`) and what `set_async(False)` / LUCID_RASTER_EXACT=1 select for everything.

Async mode (the default once a problem size has been seen `warm_calls` times) sizes the binning buffer from the high-water
mark of the instance counts seen so far for the same (device, P, H, W) times `headroom`; every kernel takes its counts from
the device-side header, and a view that needs more instances than the buffer holds is flagged there (`overflow`), never
written out of bounds.  What is done about such a view is the `on_overflow` policy -- in NO case do gradients of a truncated
instance list reach the caller: the backward kernels of an overflowed view write nothing (device-side test of the same flag).

  "verify" (default)  The forward enqueues ALL its kernels and then waits for a copy of its own header that the library posts
                      right after the compaction scan (lr_request_early_header): the wait covers the preprocess and scan
                      kernels only, the rest of the forward is already queued behind them, so the GPU does not idle.  If the
                      view overflowed it is rendered again in exact mode before anything is returned: images and gradients
                      are ALWAYS those of a complete render -- for training loops and for render-only loops alike (the
                      reference's video loop does not use no_grad, /root/reference/luciddreamer.py:250-255).  The host can
                      be at most one forward ahead of the GPU.
  "drop"              Never waits -- for callers that keep several views in flight (parallel.ViewStreams selects it for its
                      own duration).  A non-blocking copy of the header follows the forward and is examined on a later call;
                      an overflowed view's image was incomplete and its gradients are zero (device-side guard); the deferred
                      check warns and raises the mark.
  "raise"             Like "drop", but the deferred check raises RuntimeError on a later call or at drain().
  "recover"           What parallel.ViewStreams runs its views under: never waits while the views are being issued (like
                      "drop"), but EVERY view's header is examined at end_step() and a view that overflowed -- its
                      gradients were zero -- is rendered and differentiated again in exact mode before the step returns, so
                      the step's gradients are those of all its views (`recovered_views` counts them).
"""
import os
import threading
import warnings


_async = os.environ.get("LUCID_RASTER_EXACT", "0") != "1"
_fused_accumulate = False
_headroom = 1.3
_hwm = {}            # (device, P, H, W) -> largest instance count observed
_pending = []        # "drop" / "raise": [[ticket of lr_header_post, key, policy]]
_CHECK_EVERY = 1      # "drop" / "raise": every k-th async forward gets its header copied back and checked
_calls = 0
_warm_calls = 2
_seen = {}           # key -> forwards seen
_on_overflow = "verify"
_tls = threading.local()     # per-thread stack of temporary policies (overflow_policy) and the force-exact depth
dropped_views = 0    # views whose gradients were zeroed by the device-side guard and NOT made up for (policies "drop" / "raise")
rerendered_views = 0 # views rendered again in exact mode by the "verify" policy
recovered_views = 0  # views of a ViewStreams step that overflowed and were run again in exact mode at end_step()

POLICIES = ("verify", "drop", "raise", "recover")
PUBLIC_POLICIES = ("verify", "drop", "raise")      # "recover" belongs to parallel.ViewStreams (overflow_policy(..., _owner=...))


def _stack():
    if not hasattr(_tls, "override"):
        _tls.override = []
    return _tls.override


def set_async(enabled: bool, headroom: float = 1.3, check_every: int = 1, warm_calls: int = 2,
              on_overflow: str = "verify"):
    """enabled: False = exact mode for every forward.
    headroom: binning capacity = high-water mark x headroom (+ 4096).
    warm_calls: the first this-many forwards of a (device, P, H, W) run in exact mode and feed the high-water mark.
    on_overflow: "verify" (default), "drop" or "raise" -- see the module docstring.
    check_every ("drop" / "raise" only): every k-th async forward has its header copied back and examined on a later
    call; k > 1 samples (views in between can overflow unnoticed -- their gradients are still zero, never wrong -- and do
    not feed the mark)."""
    global _async, _headroom, _CHECK_EVERY, _warm_calls, _on_overflow
    if on_overflow not in PUBLIC_POLICIES:
        # "recover" needs an owner that runs a lost view again (parallel.ViewStreams); as a global policy an overflowed view
        # would silently contribute nothing
        raise ValueError(f"on_overflow must be one of {PUBLIC_POLICIES}")
    _async = bool(enabled)
    _headroom = float(headroom)
    _CHECK_EVERY = max(1, int(check_every))
    _warm_calls = max(1, int(warm_calls))
    _on_overflow = on_overflow
    if not enabled:
        drain()


class overflow_policy:
    """Context manager: a temporary policy (parallel.ViewStreams keeps views in flight and must not wait in backward)."""

    def __init__(self, policy, _owner=None):
        """_owner: the object that re-runs overflowed views -- required for "recover" (parallel.ViewStreams passes itself)."""
        if policy not in POLICIES or (policy == "recover" and _owner is None):
            raise ValueError(f"policy must be one of {PUBLIC_POLICIES}")
        self.policy = policy

    def __enter__(self):
        _stack().append(self.policy)
        return self

    def __exit__(self, *exc):
        st = _stack()
        if st:
            st.pop()
        return False


class force_exact:
    """Context manager: every forward inside runs in exact mode (the reference's host round trip), whatever the mark says."""

    def __enter__(self):
        _tls.exact = getattr(_tls, "exact", 0) + 1
        return self

    def __exit__(self, *exc):
        _tls.exact -= 1
        return False


def current_policy() -> str:
    st = _stack()
    return st[-1] if st else _on_overflow


def set_strict_parity(enabled: bool):
    """Strict evaluation of the blend (csrc/common.h gauss_weight<STRICT>): alpha from the reference's own expression in the
    reference's operand order with every operation rounded on its own and expf -- the float operations of the reference
    compiled without contraction -- instead of the Horner form with exp2 on pre-scaled coefficients.  Every discrete decision
    of the blend then falls as it does in the reference: images agree within 1e-5 on EVERY pixel (the default agrees outside
    the 0-3 pixels per 1080p view that sit within an ulp of a threshold) and no gradient row needs an exemption.  About 15 more
    instructions per pixel step in both blend kernels: a parity instrument, not the default.  Process-wide; do not change it
    between a forward and its backward."""
    from . import _lib
    _lib.tune_set("strict", 1 if enabled else -1)


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
    if overflow and policy == "recover":
        return True                       # the owner of the view (parallel.ViewStreams) runs it again
    if overflow and policy != "verify":
        dropped_views += 1
        msg = (f"luciddreamer_amd async mode: a view needed {num_instances} tile instances, more than its binning "
               "capacity; its image was incomplete and its gradients were ZERO (the backward kernels skip an "
               "overflowed view). The capacity has been raised")
        if policy == "raise":
            raise RuntimeError(msg + "; re-run the view, or use on_overflow='verify' / exact mode "
                                     "(luciddreamer_amd.config.set_async).")
        warnings.warn(msg + ".")
    return bool(overflow)


def _poll(block=False):
    """Examine the completed header copies of "drop" / "raise" forwards, in order."""
    from . import _C
    while _pending:
        entry = _pending[0]
        words = _C.header_poll(entry[0], block)
        if words is None:
            break
        _pending.pop(0)
        entry[3] = None                   # examined even if _digest raises
        entry[3] = _digest(words, entry[1], entry[2])


def drain():
    """Wait for all outstanding async forwards and surface a deferred overflow, if any."""
    _poll(block=True)


def capacity_for(means3D, rs) -> int:
    """0 = exact mode for this call; otherwise the number of tile instances to size the binning buffer for."""
    if not _async or means3D.shape[0] == 0 or getattr(_tls, "exact", 0):
        return 0
    _poll()
    key = _key(means3D, rs)
    est = _hwm.get(key)
    n = _seen.get(key, 0)
    _seen[key] = n + 1
    if est is None or (_warm_calls > 1 and n < _warm_calls):
        return 0                      # first sighting(s) of this problem size: measure exactly
    return int(est * _headroom) + 4096


def verifying(capacity) -> bool:
    """Does this async forward verify itself (policy "verify")?  Then the caller requests the early header ticket
    (_C.request_early_header) before the forward and passes the ticket to verify()."""
    return capacity != 0 and current_policy() == "verify"


def verify(means3D, rs, ticket) -> bool:
    """Policy "verify", right after an async forward was enqueued: wait for the header copy the library posted after the
    compaction scan, feed the mark, return True if the view overflowed (the caller renders it again in exact mode)."""
    global rerendered_views
    from . import _C
    if ticket is None or ticket < 0:
        return False
    over = _digest(_C.header_poll(ticket, True), _key(means3D, rs), "verify")
    if over:
        rerendered_views += 1
    return over


def offer_grad_output(grad_output):
    """parallel.ViewStreams.run_view: the NEXT rasterizer call on this thread may run its backward right behind its forward
    with this dL/dcolor (one call into the binding, no autograd node: rasterizer.rasterize_gaussians); None withdraws the
    offer.  Whether it was taken: grad_output_taken()."""
    _tls.offered_grad = grad_output
    _tls.grad_taken = False


def offered_grad_output():
    return getattr(_tls, "offered_grad", None)


def mark_grad_output_taken():
    _tls.offered_grad = None
    _tls.grad_taken = True


def grad_output_taken() -> bool:
    """True once if the offer of offer_grad_output() was taken (the view's gradients are already accumulated); clears it."""
    taken, _tls.grad_taken, _tls.offered_grad = getattr(_tls, "grad_taken", False), False, None
    return taken


def take_last_entry():
    """The deferred-check entry of the LAST forward issued by this thread ([ticket, key, policy, overflowed]), or None if that
    forward posted none (exact mode, "verify", a sampled-out view); cleared by the call.  parallel.ViewStreams ties a view to
    its header with this -- the length of `_pending` is no guide: the forward itself polls and retires older entries."""
    entry, _tls.last_entry = getattr(_tls, "last_entry", None), None
    return entry


def note_forward(means3D, rs, num_rendered, geom, capacity):
    """After a forward.  Exact forwards feed the mark directly; async ones under "drop" / "raise" post the header copy that
    the deferred check examines ("verify" has looked at its own already)."""
    _tls.last_entry = None
    if not _async or means3D.shape[0] == 0:
        return
    key = _key(means3D, rs)
    if capacity == 0:
        if num_rendered > _hwm.get(key, 0):
            _hwm[key] = num_rendered
        return
    policy = current_policy()
    if policy == "verify":
        return
    global _calls
    _calls += 1
    if policy != "recover" and _calls % _CHECK_EVERY:
        return
    from . import _C
    # the forward left its header in the library's forward log (host-visible memory written by the scan kernel: no copy, no
    # event, nothing enqueued); a library without one falls back to a 48-byte copy + event (lr_header_post).
    # Entry: [ticket, key, policy, overflowed (None until examined)]
    ticket = _C.last_forward_ticket() if hasattr(_C, "last_forward_ticket") else -1
    entry = [ticket if ticket >= 0 else _C.header_post(geom), key, policy, None]
    _pending.append(entry)
    _tls.last_entry = entry
