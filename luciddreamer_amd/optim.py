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


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        # one launch per (betas, eps, step count) combination -- a single one for a GaussianModel
        batches = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdam needs float32 parameters and gradients on a HIP device")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)     # a host tensor, as torch.optim.Adam keeps it
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not torch.is_tensor(st["step"]):                             # state loaded from an older checkpoint
                    st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
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
                with torch.cuda.device(dev):
                    rc = L.lr_adam_step(n, arr(0), arr(1), arr(2), arr(3), numel, lrs, float(b1), float(b2), float(eps),
                                        int(step), torch.cuda.current_stream(dev).cuda_stream)
                if rc < 0:
                    _lib.raise_for(rc, "lr_adam_step")
        return loss
