"""Drop-in replacement of the reference's pybind11 module `depth_diff_gaussian_rasterization_min._C`.

Same three functions, same positional arguments, same return tuples:
    rasterize_gaussians           RAST/rasterize_points.h:18-38,  RAST/rasterize_points.cu:35-117
    rasterize_gaussians_backward  RAST/rasterize_points.h:40-63,  RAST/rasterize_points.cu:119-200
    mark_visible                  RAST/rasterize_points.h:65-68,  RAST/rasterize_points.cu:202-221
Tensors are torch tensors on a HIP device; the work is done by liblucid_raster.so through its
C-ABI (include/lucid_raster.h) on the CURRENT torch stream.  torch is used here only for device
memory and streams.

Beyond the reference's contract (all optional, keyword-only):
    binning_capacity : > 0 runs lr_forward in async mode (no host sync; see lucid_raster.h)
"""
import threading

import torch

from . import _lib

NUM_CHANNELS = 3
_tls = threading.local()
# lr_backward accumulate_mask bit per gradient output (LR_ACC_* in include/lucid_raster.h)
ACC_BITS = {"means2D": 0, "opacity": 2, "colors": 3, "means3D": 4, "cov3D": 5, "sh": 6, "scales": 7, "rotations": 8}


def _alloc_cb(nbytes, user):
    """lr_alloc_fn: allocate a torch uint8 tensor on the call's device and remember it in slot `user`."""
    call = _tls.call
    t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=call["device"])
    call["bufs"][int(user or 0)] = t
    return t.data_ptr()


_ALLOC = _lib.ALLOC_FN(_alloc_cb)


def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"luciddreamer_amd: {name} must be on a HIP device (got {t.device}); this rasterizer has no CPU path "
            "(neither has the reference: RAST/rasterize_points.cu:72)")


def _f32(t, device, name):
    """float32, contiguous, on `device`; empty tensors (the reference's `torch.Tensor([])` placeholders,
    RAST/.../__init__.py:198-208) become None."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():
        return t                                            # the common case: nothing to do
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs ~5 us)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, *, binning_capacity=0):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")          # rasterize_points.cu:57-59
    _require_device(means3D, "means3D")
    dev = means3D.device
    L = _lib.lib()
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)

    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    if P == 0:
        # rasterize_points.cu:68-82: zero images, empty scratch, nothing launched
        empty = torch.empty((0,), dtype=torch.uint8, device=dev)
        return 0, out_color.zero_(), out_depth.zero_(), radii, empty, empty.clone(), empty.clone()

    means3D_c = _f32(means3D, dev, "means3D")
    bg = _f32(background, dev, "background")
    colors_c = _f32(colors, dev, "colors_precomp")
    opacity_c = _f32(opacity, dev, "opacities")
    scales_c = _f32(scales, dev, "scales")
    rot_c = _f32(rotations, dev, "rotations")
    cov_c = _f32(cov3D_precomp, dev, "cov3D_precomp")
    view = _f32(viewmatrix, dev, "viewmatrix")
    proj = _f32(projmatrix, dev, "projmatrix")
    cam = _f32(campos, dev, "campos")
    sh_c = _f32(sh, dev, "sh")
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0 and sh.size(0) != 0) else 0  # rasterize_points.cu:84-88

    call = {"device": dev, "bufs": [None, None, None]}
    _tls.call = call
    try:
        with _on_device(dev):
            rc = L.lr_forward(_ALLOC, 0, _ALLOC, 1, _ALLOC, 2, P, int(degree), M, _ptr(bg), W, H,
                              _ptr(means3D_c), _ptr(sh_c), _ptr(colors_c), _ptr(opacity_c), _ptr(scales_c),
                              float(scale_modifier), _ptr(rot_c), _ptr(cov_c), _ptr(view), _ptr(proj), _ptr(cam),
                              float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                              out_color.data_ptr(), out_depth.data_ptr(), radii.data_ptr(), int(bool(debug)),
                              int(binning_capacity), _stream(dev))
    finally:
        _tls.call = None
    if rc < 0 and rc != _lib.LR_NUM_RENDERED_ON_DEVICE:
        _lib.raise_for(rc, "rasterize_gaussians")
    geom, binning, img = call["bufs"]
    if binning is None:
        binning = torch.empty((0,), dtype=torch.uint8, device=dev)
    return rc, out_color, out_depth, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 debug, *, binning_capacity=0, accumulate_into=None, skip_unused=False):
    """accumulate_into (optional): {name: tensor} with names among ACC_BITS; the gradient of that input is
    ADDED in place into the given contiguous float32 tensor (rows of culled Gaussians untouched) and the
    corresponding slot of the returned tuple is None.
    skip_unused: do not materialise gradients of inputs that are absent (dL_dcolors when SHs are used, dL_dcov3D /
    dL_dscales / dL_drotations for the representation not in use); their slots are None.  The reference always
    returns all eight tensors, so the default keeps that."""
    _require_device(means3D, "means3D")
    dev = means3D.device
    L = _lib.lib()
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0 and sh.size(0) != 0) else 0

    # Outputs are fully written by the library (culled rows = 0): no zero-fill (cf. rasterize_points.cu:154-162).
    opt = dict(dtype=torch.float32, device=dev)
    acc = accumulate_into or {}
    mask = 0
    shapes = {"means3D": (P, 3), "means2D": (P, 3), "colors": (P, NUM_CHANNELS), "opacity": (P, 1), "cov3D": (P, 6),
              "sh": (P, M, 3), "scales": (P, 3), "rotations": (P, 4)}
    outs = {}
    unused = set()
    if skip_unused:
        if colors is None or colors.numel() == 0:
            unused.add("colors")
        if cov3D_precomp is None or cov3D_precomp.numel() == 0:
            unused.add("cov3D")
        if scales is None or scales.numel() == 0:
            unused.update(("scales", "rotations"))
    for name, shape in shapes.items():
        if name in unused:
            outs[name] = None
            continue
        t = acc.get(name)
        if t is not None:
            n = 1
            for d in shape:
                n *= d
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or t.numel() != n:
                raise RuntimeError(f"accumulate_into[{name!r}] must be a contiguous float32 tensor of {n} elements on {dev}")
            mask |= 1 << ACC_BITS[name]
            outs[name] = t
        else:
            outs[name] = torch.empty(shape, **opt)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity = outs["means3D"], outs["means2D"], outs["colors"], outs["opacity"]
    dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = outs["cov3D"], outs["sh"], outs["scales"], outs["rotations"]
    if P != 0:
        means3D_c = _f32(means3D, dev, "means3D")
        bg = _f32(background, dev, "background")
        colors_c = _f32(colors, dev, "colors_precomp")
        scales_c = _f32(scales, dev, "scales")
        rot_c = _f32(rotations, dev, "rotations")
        cov_c = _f32(cov3D_precomp, dev, "cov3D_precomp")
        view = _f32(viewmatrix, dev, "viewmatrix")
        proj = _f32(projmatrix, dev, "projmatrix")
        cam = _f32(campos, dev, "campos")
        sh_c = _f32(sh, dev, "sh")
        g_color = _f32(dL_dout_color, dev, "dL_dout_color")
        g_depth = _f32(dL_dout_depth, dev, "dL_dout_depth") if dL_dout_depth is not None else None
        radii_c = radii.contiguous()
        with _on_device(dev):
            rc = L.lr_backward(P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(means3D_c), _ptr(sh_c),
                               _ptr(colors_c), _ptr(scales_c), float(scale_modifier), _ptr(rot_c), _ptr(cov_c),
                               _ptr(view), _ptr(proj), _ptr(cam), float(tan_fovx), float(tan_fovy),
                               radii_c.data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                               imageBuffer.data_ptr(), _ptr(g_color), _ptr(g_depth),
                               dL_dmeans2D.data_ptr(), None, dL_dopacity.data_ptr(), _ptr(dL_dcolors),
                               dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D), _ptr(dL_dsh) if M else None,
                               _ptr(dL_dscales), _ptr(dL_drotations), int(bool(debug)),
                               int(binning_capacity), mask, _stream(dev))
        if rc < 0:
            _lib.raise_for(rc, "rasterize_gaussians_backward")
    r = lambda name: None if name in acc and acc[name] is not None else outs[name]
    return (r("means2D"), r("colors"), r("opacity"), r("means3D"), r("cov3D"), r("sh"), r("scales"), r("rotations"))


def rasterize_gaussians_raw(background, xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                            degree, campos, debug, *, binning_capacity=0):
    """Forward on the STORED GaussianModel tensors (lr_forward_raw): exp / normalize / sigmoid and
    the features_dc|features_rest split are handled inside the kernels.  Returns the tuple of rasterize_gaussians."""
    if xyz.ndimension() != 2 or xyz.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_device(xyz, "xyz")
    dev = xyz.device
    L = _lib.lib()
    P, H, W = int(xyz.size(0)), int(image_height), int(image_width)
    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    if P == 0:
        empty = torch.empty((0,), dtype=torch.uint8, device=dev)
        return 0, out_color.zero_(), out_depth.zero_(), radii, empty, empty.clone(), empty.clone()
    if features_dc.numel() != 3 * P:
        raise RuntimeError("features_dc must have dimensions (num_points, 1, 3)")
    M = 1 + (int(features_rest.size(1)) if (features_rest is not None and features_rest.numel() != 0) else 0)
    t = [_f32(v, dev, n) for v, n in ((background, "background"), (xyz, "xyz"), (features_dc, "features_dc"),
                                      (features_rest, "features_rest"), (opacity_raw, "opacity"),
                                      (scaling_raw, "scaling"), (rotation_raw, "rotation"), (viewmatrix, "viewmatrix"),
                                      (projmatrix, "projmatrix"), (campos, "campos"))]
    bg, xyz_c, dc, rest, op, sc, rot, view, proj, cam = t
    call = {"device": dev, "bufs": [None, None, None]}
    _tls.call = call
    try:
        with _on_device(dev):
            rc = L.lr_forward_raw(_ALLOC, 0, _ALLOC, 1, _ALLOC, 2, P, int(degree), M, _ptr(bg), W, H, _ptr(xyz_c), _ptr(dc),
                                  _ptr(rest), _ptr(op), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(view), _ptr(proj),
                                  _ptr(cam), float(tan_fovx), float(tan_fovy), out_color.data_ptr(), out_depth.data_ptr(),
                                  radii.data_ptr(), int(bool(debug)), int(binning_capacity), _stream(dev))
    finally:
        _tls.call = None
    if rc < 0 and rc != _lib.LR_NUM_RENDERED_ON_DEVICE:
        _lib.raise_for(rc, "rasterize_gaussians_raw")
    geom, binning, img = call["bufs"]
    if binning is None:
        binning = torch.empty((0,), dtype=torch.uint8, device=dev)
    return rc, out_color, out_depth, radii, geom, binning, img


# accumulate_into names of the raw backward -> LR_ACC_* bit (features_dc and features_rest share LR_ACC_SH)
RAW_ACC_BITS = {"means2D": 0, "opacity": 2, "xyz": 4, "features": 6, "scaling": 7, "rotation": 8}


def rasterize_gaussians_raw_backward(background, xyz, radii, features_dc, features_rest, opacity_raw, scaling_raw,
                                     rotation_raw, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                     dL_dout_color, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, *,
                                     binning_capacity=0, accumulate_into=None):
    """Gradients w.r.t. the stored tensors: (means2D, xyz, features_dc, features_rest, opacity, scaling, rotation).
    accumulate_into: {"means2D","xyz","opacity","scaling","rotation": tensor, "features": (dc_grad, rest_grad)} adds
    in place (slot returned as None)."""
    _require_device(xyz, "xyz")
    dev = xyz.device
    L = _lib.lib()
    P = int(xyz.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    nrest = int(features_rest.size(1)) if (features_rest is not None and features_rest.numel() != 0) else 0
    M = 1 + nrest
    opt = dict(dtype=torch.float32, device=dev)
    acc = accumulate_into or {}
    mask = 0
    shapes = {"means2D": (P, 3), "xyz": (P, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    outs = {}

    def usable(t, n):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or t.numel() != n:
            raise RuntimeError(f"accumulate_into tensors must be contiguous float32 of the gradient's size on {dev}")
        return t
    for name, shape in shapes.items():
        t = acc.get(name)
        if t is not None:
            outs[name] = usable(t, shape[0] * shape[1])
            mask |= 1 << RAW_ACC_BITS[name]
        else:
            outs[name] = torch.empty(shape, **opt)
    fa = acc.get("features")
    if fa is not None:
        g_dc, g_rest = usable(fa[0], 3 * P), (usable(fa[1], 3 * nrest * P) if nrest else None)
        mask |= 1 << RAW_ACC_BITS["features"]
    else:
        g_dc = torch.empty((P, 1, 3), **opt)
        g_rest = torch.empty((P, nrest, 3), **opt)
    if P != 0:
        t = [_f32(v, dev, n) for v, n in ((background, "background"), (xyz, "xyz"), (features_dc, "features_dc"),
                                          (features_rest, "features_rest"), (opacity_raw, "opacity"),
                                          (scaling_raw, "scaling"), (rotation_raw, "rotation"), (viewmatrix, "viewmatrix"),
                                          (projmatrix, "projmatrix"), (campos, "campos"), (dL_dout_color, "dL_dout_color"))]
        bg, xyz_c, dc, rest, op, sc, rot, view, proj, cam, g_color = t
        radii_c = radii.contiguous()
        with _on_device(dev):
            rc = L.lr_backward_raw(P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(xyz_c), _ptr(dc), _ptr(rest), _ptr(op),
                                   _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(view), _ptr(proj), _ptr(cam),
                                   float(tan_fovx), float(tan_fovy), radii_c.data_ptr(), geomBuffer.data_ptr(),
                                   binningBuffer.data_ptr(), imageBuffer.data_ptr(), _ptr(g_color),
                                   outs["means2D"].data_ptr(), outs["opacity"].data_ptr(), outs["xyz"].data_ptr(),
                                   g_dc.data_ptr(), _ptr(g_rest) if nrest else None, outs["scaling"].data_ptr(),
                                   outs["rotation"].data_ptr(), int(bool(debug)), int(binning_capacity), mask, _stream(dev))
        if rc < 0:
            _lib.raise_for(rc, "rasterize_gaussians_raw_backward")
    r = lambda name: None if acc.get(name) is not None else outs[name]
    f = (None, None) if fa is not None else (g_dc, g_rest)
    return (r("means2D"), r("xyz"), f[0], f[1], r("opacity"), r("scaling"), r("rotation"))


def mark_visible(means3D, viewmatrix, projmatrix):
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.empty((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m = _f32(means3D, dev, "means3D")
        v = _f32(viewmatrix, dev, "viewmatrix")
        p = _f32(projmatrix, dev, "projmatrix")
        with _on_device(dev):
            rc = _lib.lib().lr_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), present.data_ptr(), _stream(dev))
        if rc < 0:
            _lib.raise_for(rc, "mark_visible")
    return present


def check(geomBuffer):
    """Synchronise and return num_rendered of a forward; raises on async-mode overflow / prefiltered trap."""
    import ctypes
    n = ctypes.c_longlong(0)
    dev = geomBuffer.device
    with _on_device(dev):
        rc = _lib.lib().lr_check(geomBuffer.data_ptr(), ctypes.byref(n), _stream(dev))
    if rc < 0:
        _lib.raise_for(rc, "check")
    return int(n.value)
