"""FusedAdam: torch.optim.Adam's interface and arithmetic, one HIP launch per step (lr_adam_step).

Drop-in for `torch.optim.Adam(l, lr=0.0, eps=1e-15)` in GaussianModel.training_setup
(/root/reference/scene/gaussian_model.py:152-165): same param_groups (one tensor and one learning rate per group, named),
same state layout (`state[p]["step"]`, `["exp_avg"]`, `["exp_avg_sq"]`), so update_learning_rate, the densification
code (reference's or luciddreamer_amd.densify) and checkpoints keep working.  No weight decay, amsgrad or maximize
(the reference uses none).  Requires float32 parameters on a HIP device; no CPU path.
"""
import ctypes

import torch

from . import _lib

# Group names of GaussianModel.training_setup (R/scene/gaussian_model.py:155-162) in the tensor order of lr_backward_raw_adam
FUSED_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_armed = None                # the FusedAdam whose step the NEXT raw-mode rasterizer backward takes (arm_fused_backward)


def take_armed(tensors):
    """rasterizer._RasterizeGaussiansRaw.backward: the armed optimizer if its six parameters ARE `tensors` (identity, in
    FUSED_ORDER), else None.  Taking disarms."""
    global _armed
    opt, _armed = _armed, None
    if opt is None:
        return None
    params = opt._fused_params()
    if params is None or len(params) != len(tensors) or any(a is not b for a, b in zip(params, tensors)):
        return None
    return opt


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._fused_pending = None       # the parameters an armed backward has already stepped, until step() has checked the iteration

    # ---- the step taken by the backward pass (lr_backward_raw_adam) ------------------------------------------
    def _fused_groups(self):
        by_name = {g.get("name"): g for g in self.param_groups}
        if any(n not in by_name or len(by_name[n]["params"]) != 1 for n in FUSED_ORDER):
            return None
        groups = [by_name[n] for n in FUSED_ORDER]
        if len({(g["betas"][0], g["betas"][1], g["eps"]) for g in groups}) != 1:
            return None
        return groups

    def _fused_params(self):
        groups = self._fused_groups()
        return None if groups is None else [g["params"][0] for g in groups]

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)         # a host tensor, as torch.optim.Adam keeps it
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not torch.is_tensor(st["step"]):                                 # state loaded from an older checkpoint
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        return st

    def arm_fused_backward(self) -> bool:
        """The NEXT raw-mode rasterizer backward over exactly this optimizer's six GaussianModel tensors (groups named
        xyz / f_dc / f_rest / opacity / scaling / rotation, one tensor each) hands its gradients to this optimizer PRIVATELY:
        it writes only the rows of the Gaussians the view visits into tensors the optimizer keeps until step() (no zero-fill of the other rows:
        LR_ACC_NO_ZERO_FILL), returns no gradient to autograd -- param.grad stays None -- and launches lr_adam_step_masked right
        behind its own kernels (apply_armed_step: the step takes the gradient of an unvisited Gaussian as zero without reading
        it); the step() that follows only checks the iteration.  Gone per iteration:
        the zero-fill pass over 236 B per Gaussian, the read of those zeros, the allocation of six gradient tensors.
        Parameters and moments after the pair are bit-identical to backward + step().  For a loop in which EVERY armed backward
        is followed by exactly one step() with no other gradient source and no change of the parameter set in between
        (R/luciddreamer.py:296-327 on iterations that neither densify nor reset opacities:
        luciddreamer_amd.install(..., fuse_step=True) arms from update_learning_rate by that schedule).  What the library can
        see of a deviation is refused loudly: step() raises if a parameter was replaced or received a .grad meanwhile.
        Returns False (nothing armed) when the groups do not have that shape.
        (Round 6 first took the step INSIDE the per-Gaussian backward kernel: the same bits, no faster -- the step's 24 B per
        element are moved at 1.6-3.9 TB/s there against 5 TB/s by a streaming kernel; profiles/r06n_*.)"""
        global _armed
        if self._fused_pending is not None:
            raise RuntimeError("FusedAdam.arm_fused_backward: the previous fused backward has not been finished by step()")
        params = self._fused_params()
        if params is None or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
            _armed = None
            return False
        _armed = self
        return True

    def disarm(self):
        global _armed
        if _armed is self:
            _armed = None

    def apply_armed_step(self, geom, params, grads):
        """Called by the rasterizer's backward right after it has enqueued the armed backward's kernels: the masked step is
        launched HERE, from inside loss.backward(), not when the loop reaches optimizer.step() -- on a LucidDreamer-sized view the
        host needs ~130 us for the loop's own lines between the two (the max-radii update, the densification statistics, the
        step's Python), during which the GPU would wait for the 290 us Adam kernel to arrive; step() then only checks that the
        iteration went as an armed iteration must.  Nothing in between reads the parameters (the statistics read dL/dmeans2D and
        the radii), and iterations that densify or reset opacities are never armed."""
        from . import _C
        groups = self._fused_groups()
        states = [self._state_of(p) for p in params]
        steps = {int(st["step"].item()) for st in states}
        if len(steps) != 1:
            raise RuntimeError("FusedAdam: the six tensors have different step counts; the masked step needs one")
        for st in states:
            st["step"] += 1
        b1, b2 = groups[0]["betas"]
        with _lib.on_device(params[0].device), torch.no_grad():
            _C.adam_step_masked(list(params), list(grads), [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states],
                                [float(g["lr"]) for g in groups], float(b1), float(b2), float(groups[0]["eps"]), steps.pop() + 1, geom)
        self._fused_pending = list(params)

    def _finish_fused(self):
        params = self._fused_pending
        self._fused_pending = None
        now = self._fused_params()
        if now is None or any(a is not b for a, b in zip(now, params)):
            raise RuntimeError("FusedAdam.step: the parameter set changed between an armed backward and step() -- the step has "
                               "already been taken on the old tensors; do not arm iterations that densify / prune / replace tensors")
        if any(p.grad is not None for p in params):
            raise RuntimeError("FusedAdam.step: a parameter received a .grad after an armed backward had taken this iteration's step "
                               "(a second backward, or another loss term): that gradient cannot be applied any more; do not arm "
                               "such iterations")
        return set(id(p) for p in params)

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad; the common case of the armed loop -- every .grad already None -- costs one pass over
        the groups instead of the base class's bookkeeping (17 us per iteration on a path whose host time after the forward is
        what a LucidDreamer-sized iteration lasts)."""
        if set_to_none:
            clean = True
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        clean = False
                        break
                if not clean:
                    break
            if clean:
                return
        return super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        self.disarm()                        # an armed backward that never ran (no loss.backward() this iteration)
        done = self._finish_fused() if self._fused_pending is not None else ()
        if done and len(done) == sum(len(g["params"]) for g in self.param_groups):
            return loss                      # the armed pair covered every parameter of this optimizer
        # one launch per (betas, eps, step count) combination -- a single one for a GaussianModel
        batches = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None or id(p) in done:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdam needs float32 parameters and gradients on a HIP device")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self._state_of(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("FusedAdam needs contiguous parameters and moments")
                key = (group["betas"][0], group["betas"][1], group["eps"], int(st["step"].item()), p.device)
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (b1, b2, eps, step, dev), items in batches.items():
            for i in range(0, len(items), 16):
                chunk = items[i:i + 16]
                n = len(chunk)
                arr = lambda k: (ctypes.c_void_p * n)(*[it[k].data_ptr() for it in chunk])
                numel = (ctypes.c_ulonglong * n)(*[it[0].numel() for it in chunk])
                lrs = (ctypes.c_double * n)(*[it[4] for it in chunk])
                with _lib.on_device(dev):
                    rc = L.lr_adam_step(n, arr(0), arr(1), arr(2), arr(3), numel, lrs, float(b1), float(b2), float(eps),
                                        int(step), torch.cuda.current_stream(dev).cuda_stream)
                if rc < 0:
                    _lib.raise_for(rc, "lr_adam_step")
        return loss
