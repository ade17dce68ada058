"""Fused photometric loss of the training loop on the MI355X.

    loss = l1_dssim_loss(image, gt, lambda_dssim)      # == (1-l)*l1_loss(image, gt) + l*(1 - ssim(image, gt))

is what /root/reference/luciddreamer.py:301-304 computes with utils/loss.py's l1_loss (:18-19) and ssim (:37-69).
`l1_loss` and `ssim` with the reference's names and meaning are provided as well.  One HIP kernel pass forward and one
backward (luciddreamer_amd/csrc/loss.hip) through the C-ABI (lr_l1_dssim_forward / lr_l1_dssim_backward); no CPU or
PyTorch fallback.  Gradients flow to `image` only (the target is data).
"""
import torch

from . import _lib

_WS_BYTES = {}                      # (C, H, W) -> lr_loss_workspace_bytes: a C call per forward otherwise


def _ws_bytes(L, C, H, W):
    n = _WS_BYTES.get((C, H, W))
    if n is None:
        n = _WS_BYTES[(C, H, W)] = int(L.lr_loss_workspace_bytes(C, H, W))
    return n


def _weight(t, dev):
    """An upstream gradient as the one-float device tensor the kernels read: as it is when autograd already hands over a float32
    scalar on the device (the usual case: three tensor ops saved per backward), converted otherwise."""
    if t is None:
        return torch.zeros(1, device=dev)
    if t.dtype is torch.float32 and t.device == dev and t.numel() == 1 and t.is_contiguous():
        return t
    return t.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()


class _L1DSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("luciddreamer_amd.loss: image and gt must be on a HIP device (no CPU path)")
        if image.shape != gt.shape or image.dim() < 2:
            raise RuntimeError(f"image {tuple(image.shape)} and gt {tuple(gt.shape)} must have the same [..., H, W] shape")
        if image.dtype != torch.float32 or gt.dtype != torch.float32:
            raise RuntimeError("image and gt must be float32")
        x, g = image.contiguous(), gt.contiguous()
        H, W = int(x.shape[-2]), int(x.shape[-1])
        C = x.numel() // (H * W)
        L = _lib.lib()
        dev = x.device
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        ws = torch.empty((_ws_bytes(L, C, H, W),), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = L.lr_l1_dssim_forward(C, H, W, x.data_ptr(), g.data_ptr(), float(lambda_dssim), out3.data_ptr(),
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "l1_dssim_loss")
        ctx.save_for_backward(x, g, ws)
        ctx.lam, ctx.dims, ctx.in_shape = float(lambda_dssim), (C, H, W), image.shape
        ctx.parts = out3                      # {loss, l1, ssim}, device
        return out3[0]

    @staticmethod
    def backward(ctx, grad_out):
        x, g, ws = ctx.saved_tensors
        C, H, W = ctx.dims
        L = _lib.lib()
        dev = x.device
        up = _weight(grad_out, dev)
        grad = torch.empty_like(x)
        with _lib.on_device(dev):
            rc = L.lr_l1_dssim_backward(C, H, W, x.data_ptr(), g.data_ptr(), ctx.lam, up.data_ptr(), ws.data_ptr(),
                                        grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "l1_dssim_loss backward")
        return grad.view(ctx.in_shape), None, None


def l1_dssim_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) * mean|image - gt| + lambda * (1 - SSIM(image, gt)); image, gt: [C,H,W] (or [B,C,H,W])."""
    return _L1DSSIM.apply(image, gt, lambda_dssim)


def l1_loss(network_output, gt):
    """utils/loss.py:18-19."""
    return _L1DSSIM.apply(network_output, gt, 0.0)


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss.py:37-46 (window 11, averaged over everything -- the only configuration the training loop uses)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("fused ssim supports window_size=11, size_average=True")
    return 1.0 - _L1DSSIM.apply(img1, img2, 1.0)


class _L1SSIMPair(torch.autograd.Function):
    """(l1, ssim) of one image pair from ONE forward pass, for callers that weight the two means themselves
    (/root/reference/luciddreamer.py:301-303: `(1 - l) * l1_loss(image, gt) + l * (1 - ssim(image, gt))`); the backward takes
    both grad_outputs as device scalars (lr_l1_dssim_backward_weights)."""

    @staticmethod
    def forward(ctx, image, gt):
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("luciddreamer_amd.loss: image and gt must be on a HIP device (no CPU path)")
        if image.shape != gt.shape or image.dim() < 2 or image.dtype != torch.float32 or gt.dtype != torch.float32:
            raise RuntimeError("image and gt must be float32 tensors of the same [..., H, W] shape")
        x, g = image.contiguous(), gt.contiguous()
        H, W = int(x.shape[-2]), int(x.shape[-1])
        C = x.numel() // (H * W)
        L = _lib.lib()
        dev = x.device
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        ws = torch.empty((_ws_bytes(L, C, H, W),), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = L.lr_l1_dssim_forward(C, H, W, x.data_ptr(), g.data_ptr(), 0.0, out3.data_ptr(), ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "l1 / ssim pair")
        ctx.save_for_backward(x, g, ws)
        ctx.dims, ctx.in_shape = (C, H, W), image.shape
        return out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        x, g, ws = ctx.saved_tensors
        C, H, W = ctx.dims
        L = _lib.lib()
        dev = x.device
        w1, w2 = _weight(g_l1, dev), _weight(g_ssim, dev)
        grad = torch.empty_like(x)
        with _lib.on_device(dev):
            rc = L.lr_l1_dssim_backward_weights(C, H, W, x.data_ptr(), g.data_ptr(), w1.data_ptr(), w2.data_ptr(), ws.data_ptr(),
                                                grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "l1 / ssim pair backward")
        return grad.view(ctx.in_shape), None


class PairedLoss:
    """`l1_loss` and `ssim` with the reference's signatures that share ONE kernel pass when they are called, in either order,
    on the same (image, gt) tensors -- which is what the training loop does.  luciddreamer_amd.install() puts a pair of these
    in place of utils/loss.py's functions.  A call with other tensors (or ssim with another window) computes on its own."""

    def __init__(self, fallback_l1=None, fallback_ssim=None):
        """fallback_*: what a call the shared pass does not cover is handed to (install() passes the functions it replaces, so
        the caller's own code keeps serving other window sizes, per-image means, host tensors); without one, l1 of anything
        else is the reference's expression and such an ssim call raises."""
        self._key, self._pair = None, None
        self._fallback_l1, self._fallback_ssim = fallback_l1, fallback_ssim

    @staticmethod
    def _fusable(a, b):
        return (torch.is_tensor(a) and torch.is_tensor(b) and a.is_cuda and b.is_cuda and a.shape == b.shape and a.dim() >= 3
                and a.shape[-3] == 3 and a.dtype == torch.float32 and b.dtype == torch.float32)

    def _get(self, a, b):
        # the pending pair is keyed on the tensor OBJECTS (held here until the second call or the next key: an id() alone can
        # be re-used by a later tensor at the same allocator address once the first is gone -- eval loops under no_grad) and
        # their versions
        k = self._key
        if (k is not None and self._pair is not None and k[0] is a and k[1] is b and k[2] == a._version
                and k[3] == b._version and k[4] == a.data_ptr() and k[5] == b.data_ptr()):
            pair, self._key, self._pair = self._pair, None, None       # second of the two calls: hand out and forget
            return pair
        self._key, self._pair = None, None                             # another pair: drop the graph / workspace of the old one
        pair = _L1SSIMPair.apply(a, b)
        self._key, self._pair = (a, b, a._version, b._version, a.data_ptr(), b.data_ptr()), pair
        return pair

    def l1_loss(self, network_output, gt):
        # only what the loop pairs with ssim goes through the shared pass: an RGB image against its target.  Anything else
        # (a depth map, a vector) is the reference's own expression, utils/loss.py:18-19
        if not self._fusable(network_output, gt) or network_output.dim() != 3:
            if self._fallback_l1 is not None:
                return self._fallback_l1(network_output, gt)
            return torch.abs(network_output - gt).mean()
        return self._get(network_output, gt)[0]

    def ssim(self, img1, img2, window_size=11, size_average=True):
        if window_size != 11 or not size_average or not self._fusable(img1, img2):
            if self._fallback_ssim is not None:
                return self._fallback_ssim(img1, img2, window_size, size_average)
            raise NotImplementedError("fused ssim: float32 [..., 3, H, W] device tensors, window_size=11, size_average=True")
        return self._get(img1, img2)[1]
