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
                         indexing, no host round trip); the visibility filter returned by the replaced render carries the radii.
  densify / prune / ply  densify.patch(cls): prune_points, densification_postfix, densify_and_clone / _split / _and_prune,
                         save_ply / load_ply over the library's row store.

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


_STORED = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _raw_ok(pc, opt, override_color):
    if override_color is not None or getattr(opt, "compute_cov3D_python", False) or getattr(opt, "convert_SHs_python", False):
        return False
    for a in _STORED:
        t = getattr(pc, a, None)
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return False
    return True


def install(gaussian_renderer=None, loss=None, gaussian_model=None, *, render=True, losses=True, adam=True, stats=True,
            densify=True, rebind=True):
    """gaussian_renderer, loss: the reference's modules (or None to leave alone); gaussian_model: its GaussianModel class or
    the module that defines it.  A single namespace with attributes `gaussian_renderer`, `loss`, `gaussian_model` (what
    oracle/ref_python.reference_modules yields) may be passed as the first argument.  Returns a handle for uninstall()."""
    if gaussian_renderer is not None and loss is None and gaussian_model is None and hasattr(gaussian_renderer, "gaussian_renderer"):
        ns = gaussian_renderer
        gaussian_renderer, loss, gaussian_model = ns.gaussian_renderer, getattr(ns, "loss", None), getattr(ns, "gaussian_model", None)
    cls = getattr(gaussian_model, "GaussianModel", gaussian_model)
    h = _Handle()
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
                out["visibility_filter"]._lr_radii = out["radii"]        # for the fused densification statistics below
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
        h.set(cls, "training_setup", training_setup)

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
