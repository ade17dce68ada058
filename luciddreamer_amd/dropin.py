"""One call that switches an UNCHANGED LucidDreamer caller onto the fused pieces of this library.

    import gaussian_renderer, utils.loss, scene.gaussian_model          # the reference's own modules
    import luciddreamer_amd
    luciddreamer_amd.install(gaussian_renderer, utils.loss, scene.gaussian_model.GaussianModel)

What is replaced (each one individually switchable; `uninstall(handle)` puts everything back):

  render                 gaussian_renderer.render (R/gaussian_renderer/__init__.py:18-104) -> gaussian_renderer.render_raw when
                         the call is the training configuration (SH colours from the model, scale / rotation covariance, no
                         override colour) and the model exposes its stored tensors: exp / normalize / sigmoid / cat and their
                         autograd nodes disappear into the rasterizer kernels.  Any other call goes to the original.
  l1_loss, ssim          utils/loss.py:18-69 -> one fused pass for the pair (loss.PairedLoss): R/luciddreamer.py:301-303 calls
                         them one after the other on the same tensors and weights them itself.
  Adam                   GaussianModel.training_setup (scene/gaussian_model.py:152-165) keeps building its six groups; the
                         torch.optim.Adam it ends with is replaced by optim.FusedAdam over the same groups (one launch per step).
  densification stats    GaussianModel.add_densification_stats (:405-407) -> lr_densify_stats (one kernel, no boolean-mask
                         indexing, no host round trip); the visibility filter returned by the replaced render carries the radii
                         -- and keeps the loop's own `max_radii2D[filter] = torch.max(max_radii2D[filter], radii[filter])`
                         (R/luciddreamer.py:310-312) on the device: three host round trips per iteration less.
  densify / prune / ply  densify.patch(cls): prune_points, densification_postfix, densify_and_clone / _split / _and_prune,
                         save_ply / load_ply over the library's row store.

  backward               `loss.backward()` runs on the calling thread (torch.autograd.set_multithreading_enabled(False)) while
                         installed: no hand-off to the autograd engine's device thread per iteration.

Functions the caller imported BY NAME before install() ran (`from gaussian_renderer import render`, `from utils.loss import
l1_loss, ssim` at the top of R/luciddreamer.py) are re-bound in every loaded module that holds the original object.
"""
import sys
import types

import torch


class _Handle:
    def __init__(self):
        self.undo = []

    def set(self, obj, name, value):
        had = name in vars(obj) if isinstance(obj, type) else hasattr(obj, name)
        old = vars(obj).get(name) if isinstance(obj, type) else getattr(obj, name, None)
        setattr(obj, name, value)
        self.undo.append((obj, name, old, had))


def _rebind(handle, original, replacement, skip):
    """Every loaded module that holds `original` under some global name gets `replacement` there."""
    for mod in list(sys.modules.values()):
        if not isinstance(mod, types.ModuleType) or mod in skip:
            continue
        d = getattr(mod, "__dict__", None)
        if not d:
            continue
        for k, v in list(d.items()):
            if v is original:
                handle.set(mod, k, replacement)


# ---- the visibility filter of the replaced render ---------------------------------------------------------------------------
# The caller's loop keeps `max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter])`
# (R/luciddreamer.py:310-312) in its own body: three boolean-mask indexings, each a nonzero() with a device -> host read of
# the row count -- the host stops until the iteration's whole GPU work has drained, three times per iteration (~1 ms of a
# 2.3 ms iteration at 1 M Gaussians / 512^2).  The filter this module's render returns is a bool tensor SUBCLASS: indexing a
# plain tensor with it yields a lazy row selection, torch.max / torch.maximum of two such selections stays lazy, and assigning
# the result through the same filter runs as ONE dense torch.where on the device -- same values, no row count, no host read.
# Anything else done with a lazy selection (arithmetic, in-place ops, .shape, printing ...) materialises it first and is the
# plain operation; a filter used in any other way behaves as the bool tensor it is.
lazy_assignments = 0          # masked assignments that ran without a host read (tests, diagnostics)


def _materialise(x):
    if isinstance(x, _Rows):
        return x.rows()
    if isinstance(x, _VisFilter):
        return x.as_subclass(torch.Tensor)
    if isinstance(x, (tuple, list)):
        return type(x)(_materialise(v) for v in x)
    if isinstance(x, dict):
        return {k: _materialise(v) for k, v in x.items()}
    return x


def _filter_function(func, args, kwargs):
    global lazy_assignments
    kwargs = kwargs or {}
    if not kwargs:
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], _VisFilter) and _Rows.can_select(args[0], args[1]):
            return _Rows.make("rows", args[1], args[0])
        if func in (torch.max, torch.maximum) and len(args) == 2 and all(isinstance(a, _Rows) for a in args) and args[0].mask is args[1].mask:
            return _Rows.make("max", args[0].mask, args[0], args[1])
        if func is torch.Tensor.__setitem__ and len(args) == 3 and isinstance(args[1], _VisFilter) and isinstance(args[2], _Rows) \
                and args[2].mask is args[1] and _Rows.can_select(args[0], args[1]):
            dest, m = args[0], args[1].as_subclass(torch.Tensor)
            with torch._C.DisableTorchFunction():
                dense = args[2].dense()
                if dense.shape == dest.shape:
                    if dest.dim() > 1:
                        m = m.view(-1, *([1] * (dest.dim() - 1)))
                    torch.where(m, dense if dense.dtype == dest.dtype else dense.to(dest.dtype), dest, out=dest)
                    lazy_assignments += 1
                    return None
    with torch._C.DisableTorchFunction():
        return func(*_materialise(tuple(args)), **_materialise(kwargs))


class _Rows(torch.Tensor):
    """`base[filter]` (or the elementwise max of two of them) not yet evaluated: see above."""

    @staticmethod
    def can_select(base, mask):
        return (type(base) is torch.Tensor and mask.dim() == 1 and base.dim() >= 1 and base.shape[0] == mask.shape[0]
                and base.device == mask.device and not (base.requires_grad and torch.is_grad_enabled()))

    _empty = {}                     # one zero-size tensor per device: a selection object is a fresh VIEW of it, no allocation

    @staticmethod
    def make(kind, mask, *operands):
        with torch._C.DisableTorchFunction():
            e = _Rows._empty.get(mask.device)
            if e is None:
                e = _Rows._empty[mask.device] = torch.empty(0, device=mask.device)
            r = e.as_subclass(_Rows)
        r.kind, r.mask, r.operands = kind, mask, operands
        r.versions = (mask._version,) + tuple(o._version for o in operands if kind == "rows")
        return r

    def dense(self):
        """The full-length tensor whose rows under the filter are this selection."""
        if self.kind == "rows":
            # eager indexing would have copied the rows when the selection was written down; a selection that is only
            # evaluated now must find its source as it was then -- anything else is refused, loudly
            if (self.mask._version, self.operands[0]._version) != self.versions:
                raise RuntimeError("luciddreamer_amd: a tensor indexed with the visibility filter of the installed render() was "
                                   "changed in place before the selection was used; take the selection with .clone() or "
                                   "install(..., lazy_filter=False)")
            return self.operands[0]
        return torch.maximum(self.operands[0].dense(), self.operands[1].dense())

    def rows(self):
        with torch._C.DisableTorchFunction():
            return self.dense()[self.mask.as_subclass(torch.Tensor)]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return _filter_function(func, args, kwargs)

    def __reduce_ex__(self, proto):
        return self.rows().__reduce_ex__(proto)


class _VisFilter(torch.Tensor):
    """The bool visibility filter (radii > 0) returned by the replaced render; carries the radii for the fused statistics."""

    @staticmethod
    def wrap(mask, radii):
        f = mask.as_subclass(_VisFilter)
        f._lr_radii = radii
        return f

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return _filter_function(func, args, kwargs)

    def __reduce_ex__(self, proto):
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


_STORED = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _raw_ok(pc, opt, override_color):
    if override_color is not None or getattr(opt, "compute_cov3D_python", False) or getattr(opt, "convert_SHs_python", False):
        return False
    for a in _STORED:
        t = getattr(pc, a, None)
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return False
    return True


def plain_iteration(training_args, iteration) -> bool:
    """Does iteration `iteration` of the reference's training loop (R/luciddreamer.py:283-327) run `loss.backward()` and
    `optimizer.step()` back to back on an UNCHANGED parameter set?  False on the iterations that densify / prune (:314-317),
    reset opacities (:319-322) or skip the step (:325: the last one).  training_args: the GSParams object training_setup got."""
    a = training_args
    need = ("iterations", "densify_until_iter", "densify_from_iter", "densification_interval", "opacity_reset_interval")
    if any(not hasattr(a, k) for k in need):
        return False
    if not iteration < a.iterations:
        return False
    if iteration < a.densify_until_iter:
        if iteration > a.densify_from_iter and iteration % a.densification_interval == 0:
            return False
        if iteration % a.opacity_reset_interval == 0 or (getattr(a, "white_background", False) and iteration == a.densify_from_iter):
            return False
    return True


def install(gaussian_renderer=None, loss=None, gaussian_model=None, *, render=True, losses=True, adam=True, stats=True,
            densify=True, rebind=True, lazy_filter=True, backward_on_calling_thread="auto", fuse_step=False):
    """gaussian_renderer, loss: the reference's modules (or None to leave alone); gaussian_model: its GaussianModel class or
    the module that defines it.  lazy_filter: the visibility filter the replaced render returns keeps the caller's masked
    max-radii update on the device (_VisFilter above); False = a plain bool tensor.  backward_on_calling_thread ("auto": only
    when no other Python thread exists at the time of the call; True / False force it):
    torch.autograd.set_multithreading_enabled(False), a PROCESS-WIDE setting, until uninstall() -- `loss.backward()` then runs its nodes on the thread
    that called it instead of handing them to the autograd engine's device thread and waiting: on this path (a few dozen
    short nodes per iteration, every one of them only ENQUEUES work) the hand-off and the two threads' turns at the
    interpreter lock cost more than the nodes -- the unchanged loop at 1 M Gaussians / 512^2 went 2.09 -> 1.28 ms per iteration
    on the same box (profiles/r04q_loop_segments_backward_thread.txt).  A single namespace with attributes `gaussian_renderer`, `loss`, `gaussian_model` (what
    oracle/ref_python.reference_modules yields) may be passed as the first argument.  Returns a handle for uninstall().
    fuse_step (opt-in; needs `adam` and `render`): the optimizer step of an iteration is taken BY its backward pass -- the kernel
    that has just summed a visible Gaussian's gradient applies Adam to its rows instead of storing 236 B of gradient that
    optimizer.step() would re-read and the next backward would zero-fill again; step() finishes the step for the Gaussians the
    view did not touch (optim.FusedAdam.arm_fused_backward; same bits as the unfused pair).  Armed per iteration from
    GaussianModel.update_learning_rate(iteration) -- the first call of every iteration of R/luciddreamer.py:283-327 -- on the
    iterations that loop runs backward and step() back to back on an unchanged parameter set (plain_iteration() above: not when
    it densifies, resets opacities or skips the step).  Only for that loop shape; a deviation the library can see (a parameter
    replaced, a second gradient source) raises in step()."""
    if gaussian_renderer is not None and loss is None and gaussian_model is None and hasattr(gaussian_renderer, "gaussian_renderer"):
        ns = gaussian_renderer
        gaussian_renderer, loss, gaussian_model = ns.gaussian_renderer, getattr(ns, "loss", None), getattr(ns, "gaussian_model", None)
    cls = getattr(gaussian_model, "GaussianModel", gaussian_model)
    h = _Handle()
    h.multithreading = None
    if backward_on_calling_thread == "auto":
        # PROCESS-WIDE switch (until uninstall()): right for the reference's single-threaded loop, a trap for a caller that runs
        # backward passes from threads of its own -- "auto" only takes it when this is the only Python thread at install time;
        # pass True / False to decide yourself
        import threading
        import warnings
        backward_on_calling_thread = threading.active_count() == 1
        if not backward_on_calling_thread:
            # said once, at install time: a tqdm monitor thread, a Jupyter kernel, wandb or pytest-timeout's watchdog are enough to
            # make "auto" decline, and the 2.09 -> 1.28 ms per iteration this switch is worth would be lost silently
            others = [t.name for t in threading.enumerate() if t is not threading.current_thread()]
            warnings.warn("luciddreamer_amd.install: backward_on_calling_thread='auto' found other Python threads ("
                          + ", ".join(others[:4]) + ("..." if len(others) > 4 else "") + ") and leaves the autograd engine's "
                          "threading as it is; if none of them runs backward passes, pass backward_on_calling_thread=True "
                          "(about 0.8 ms per iteration of the reference's loop at 1 M Gaussians / 512^2)", stacklevel=2)
    h.backward_on_calling_thread = bool(backward_on_calling_thread)
    if backward_on_calling_thread and hasattr(torch.autograd, "set_multithreading_enabled"):
        h.multithreading = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)
    from . import densify as dz, gaussian_renderer as gr
    from .loss import PairedLoss
    from .optim import FusedAdam

    if render and gaussian_renderer is not None:
        orig_render = gaussian_renderer.render

        def render_(viewpoint_camera, pc, opt, bg_color, scaling_modifier=1.0, override_color=None, render_only=False):
            if not _raw_ok(pc, opt, override_color):
                return orig_render(viewpoint_camera, pc, opt, bg_color, scaling_modifier, override_color, render_only)
            out = gr.render_raw(viewpoint_camera, pc, opt, bg_color, scaling_modifier, render_only)
            if not render_only:
                # carries the radii for the fused densification statistics below, and keeps the caller's own
                # `max_radii2D[filter] = torch.max(max_radii2D[filter], radii[filter])` on the device (_VisFilter above)
                if lazy_filter:
                    out["visibility_filter"] = _VisFilter.wrap(out["visibility_filter"], out["radii"])
                else:
                    out["visibility_filter"]._lr_radii = out["radii"]
            return out
        render_.__wrapped__ = orig_render
        h.set(gaussian_renderer, "render", render_)
        if rebind:
            _rebind(h, orig_render, render_, (gaussian_renderer,))

    if losses and loss is not None:
        pair = PairedLoss(fallback_l1=loss.l1_loss, fallback_ssim=loss.ssim)      # what the pair does not fuse stays the caller's
        for name in ("l1_loss", "ssim"):
            orig_fn, repl = getattr(loss, name), getattr(pair, name)
            h.set(loss, name, repl)
            if rebind:
                _rebind(h, orig_fn, repl, (loss,))

    if cls is not None and adam:
        setup = cls.training_setup

        def training_setup(self, training_args):
            setup(self, training_args)
            opt = self.optimizer
            if isinstance(opt, torch.optim.Adam) and all(p.is_cuda for g in opt.param_groups for p in g["params"]):
                groups = [{k: v for k, v in g.items() if k in ("params", "lr", "name")} for g in opt.param_groups]
                self.optimizer = FusedAdam(groups, lr=opt.defaults["lr"], betas=opt.defaults["betas"], eps=opt.defaults["eps"])
            self._lr_training_args = training_args            # the loop's schedule: what fuse_step arms by
        h.set(cls, "training_setup", training_setup)

        if fuse_step and render and hasattr(cls, "update_learning_rate"):
            update_lr = cls.update_learning_rate

            def update_learning_rate(self, iteration):
                out = update_lr(self, iteration)
                opt, args = getattr(self, "optimizer", None), getattr(self, "_lr_training_args", None)
                if isinstance(opt, FusedAdam):
                    if args is not None and plain_iteration(args, iteration):
                        opt.arm_fused_backward()
                    else:
                        opt.disarm()
                return out
            h.set(cls, "update_learning_rate", update_learning_rate)

    if cls is not None and stats:
        add = cls.add_densification_stats

        def add_densification_stats(self, viewspace_point_tensor, update_filter):
            radii = getattr(update_filter, "_lr_radii", None)
            g = viewspace_point_tensor.grad
            ok = lambda t: torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            if radii is None or g is None or not (ok(g) and ok(self.xyz_gradient_accum) and ok(self.denom) and ok(self.max_radii2D)):
                return add(self, viewspace_point_tensor, update_filter)
            # the filter IS radii > 0 (the replaced render made it); the kernel also takes max(max_radii2D, radii) on those
            # rows, which the caller has just done itself (R/luciddreamer.py:310-311): idempotent
            dz.add_densification_stats(self, viewspace_point_tensor, radii)
        h.set(cls, "add_densification_stats", add_densification_stats)

    if cls is not None and densify:
        for fn in (dz.prune_points, dz.densification_postfix, dz.densify_and_clone, dz.densify_and_split, dz.densify_and_prune,
                   dz.save_ply):
            h.set(cls, fn.__name__, fn)
        h.set(cls, "load_ply", lambda self, path: dz.load_ply(self, path))
    return h


def uninstall(handle):
    for obj, name, old, had in reversed(handle.undo):
        if had:
            setattr(obj, name, old)
        else:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
    handle.undo = []
    if getattr(handle, "multithreading", None) is not None:
        torch.autograd.set_multithreading_enabled(handle.multithreading)
        handle.multithreading = None
