"""oracle/_ref on the GPU: the reference's own rasterizer kernels, compiled by hipcc from /root/reference
(oracle/build_ref.py build_device(), oracle/ref_shim_hip/) and driven with torch device tensors.

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY (tests/, bench.py's reference_on_device leg): a device-side checker of the
HIP path at full problem sizes -- its float atomics make gradients differ from run to run, as on CUDA -- and the
"reference's own code on this MI355X" figure next to the bench line.  Never imported by the product packages.
"""
import ctypes

import torch

from . import build_ref

_lib = None
_libs = {}


def available(contract="off"):
    return build_ref.build_device(contract=contract) is not None and torch.cuda.is_available()


def lib(contract="off"):
    """contract="off": IEEE operations in source order (the checker).  "fast": the compiler's default FMA contraction, i.e.
    the reference as its own build produces it (build_ref.build_device)."""
    global _lib
    if contract not in _libs:
        path = build_ref.build_device(contract=contract)
        if path is None:
            raise RuntimeError(f"oracle/_ref gfx950 build (contract={contract}) is missing and /root/reference is not present")
        L = ctypes.CDLL(path)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.refdev_state_new.restype = vp
        L.refdev_state_new.argtypes = []
        L.refdev_state_free.argtypes = [vp]
        L.refdev_forward.restype = ci
        L.refdev_forward.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp]
        L.refdev_backward.restype = None
        L.refdev_backward.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.refdev_sync.argtypes = []
        L.refdev_dist2.argtypes = [ci, vp, vp]
        _libs[contract] = L
        if contract == "off":
            _lib = L
    return _libs[contract]


def _p(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class Renderer:
    """One view at a time; keeps the reference's three scratch buffers between forward and backward."""

    def __init__(self, contract="off"):
        self.L = lib(contract)
        self.state = self.L.refdev_state_new()

    def __del__(self):
        if getattr(self, "state", None) and self.L is not None:
            self.L.refdev_state_free(self.state)
            self.state = None

    def forward(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, view, proj, tanfovx, tanfovy,
                H, W, sh, degree, campos, prefiltered=False, sync=True):
        dev = means3D.device
        P = means3D.shape[0]
        M = sh.shape[1] if (sh is not None and sh.numel()) else 0
        self.color = torch.empty(3, H, W, device=dev)
        self.depth = torch.empty(1, H, W, device=dev)
        self.radii = torch.empty(P, dtype=torch.int32, device=dev)
        self.args = (P, degree, M, bg, W, H, means3D, sh, colors, scales, scale_modifier, rotations, cov3D, view, proj, campos,
                     tanfovx, tanfovy)
        if sync:
            torch.cuda.synchronize()
        self.R = self.L.refdev_forward(self.state, P, degree, M, _p(bg), W, H, _p(means3D), _p(sh), _p(colors), _p(opacity),
                                       _p(scales), scale_modifier, _p(rotations), _p(cov3D), _p(view), _p(proj), _p(campos),
                                       tanfovx, tanfovy, int(prefiltered), _p(self.color), _p(self.depth), _p(self.radii))
        if sync:
            self.L.refdev_sync()
        return self.R, self.color, self.depth, self.radii

    def backward(self, grad_color, sync=True):
        """Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dconic)."""
        P, degree, M, bg, W, H, means3D, sh, colors, scales, smod, rotations, cov3D, view, proj, campos, tfx, tfy = self.args
        dev = means3D.device
        e = lambda *s: torch.empty(*s, device=dev)
        if not hasattr(self, "g") or self.g[0].shape[0] != P or self.g[5].shape[1] != max(M, 1):
            self.g = (e(P, 3), e(P, 3), e(P, 1), e(P, 3), e(P, 6), e(P, max(M, 1), 3), e(P, 3), e(P, 4), e(P, 2, 2))
            self.gdepth = torch.zeros(1, H, W, device=dev)
        m2, col, op, m3, cov, gsh, sc, rot, conic = self.g
        if sync:
            torch.cuda.synchronize()
        self.L.refdev_backward(self.state, P, degree, M, _p(bg), W, H, _p(means3D), _p(sh), _p(colors), _p(scales), smod,
                               _p(rotations), _p(cov3D), _p(view), _p(proj), _p(campos), tfx, tfy, _p(self.radii), _p(grad_color),
                               _p(self.gdepth), _p(m2), _p(conic), _p(op), _p(col), _p(m3), _p(cov), _p(gsh) if M else None,
                               _p(sc), _p(rot))
        if sync:
            self.L.refdev_sync()
        return self.g


def dist2(points):
    out = torch.zeros(points.shape[0], device=points.device)
    torch.cuda.synchronize()
    lib().refdev_dist2(points.shape[0], points.contiguous().data_ptr(), out.data_ptr())
    lib().refdev_sync()
    return out
