"""Drop-in replacement of the reference's pybind11 module `depth_diff_gaussian_rasterization_min._C`.

Same three functions, same positional arguments, same return tuples:
    rasterize_gaussians           RAST/rasterize_points.h:18-38,  RAST/rasterize_points.cu:35-117
    rasterize_gaussians_backward  RAST/rasterize_points.h:40-63,  RAST/rasterize_points.cu:119-200
    mark_visible                  RAST/rasterize_points.h:65-68,  RAST/rasterize_points.cu:202-221
Tensors are torch tensors on a HIP device; the work is done by liblucid_raster.so through its C-ABI
(include/lucid_raster.h) on the CURRENT torch stream.  The marshalling (tensor checks, output and scratch
allocation, pointers, stream) is the compiled module `luciddreamer_amd._C_ext` (csrc/torch_ext.cpp, built in-tree by
`python -m luciddreamer_amd.build`); this file only adapts keyword conveniences to its positional interface.
There is no fallback: a missing extension raises on import.

Beyond the reference's contract (all optional, keyword-only):
    binning_capacity : > 0 runs lr_forward in async mode (no host sync; see lucid_raster.h)
"""
try:
    from . import _C_ext
except ImportError as e:                                        # pragma: no cover - build problem, not a code path
    raise ImportError(
        "luciddreamer_amd: the compiled binding luciddreamer_amd/_C_ext*.so is missing or does not load "
        f"({e}); build it with `python -m luciddreamer_amd.build` (needs lib/liblucid_raster.so). "
        "There is no CPU / pure-Python fallback.") from e

from . import _lib as _lib_check
_lib_check.assert_single_copy()                                 # the binding and ctypes must be on ONE build of the library

NUM_CHANNELS = 3
# order of rasterize_gaussians_backward's result tuple (RAST/rasterize_points.cu:199) = order of `accumulate`
GRAD_ORDER = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
ACC_BITS = {"means2D": 0, "opacity": 2, "colors": 3, "means3D": 4, "cov3D": 5, "sh": 6, "scales": 7, "rotations": 8}
RAW_GRAD_ORDER = ("means2D", "xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
RAW_ACC_BITS = {"means2D": 0, "opacity": 2, "xyz": 4, "features": 6, "scaling": 7, "rotation": 8}
_NONE8 = []


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, *, binning_capacity=0):
    return _C_ext.rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                                      campos, prefiltered, debug, binning_capacity)


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
    acc = _NONE8 if not accumulate_into else [accumulate_into.get(k) for k in GRAD_ORDER]
    return tuple(_C_ext.rasterize_gaussians_backward(
        background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
        debug, binning_capacity, acc, skip_unused))


def rasterize_gaussians_raw(background, xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                            degree, campos, debug, *, binning_capacity=0):
    """Forward on the STORED GaussianModel tensors (lr_forward_raw): exp / normalize / sigmoid and
    the features_dc|features_rest split are handled inside the kernels.  Returns the tuple of rasterize_gaussians."""
    if opacity_raw is None:
        raise RuntimeError("rasterize_gaussians_raw: raw mode needs features_dc, features_rest (M > 1), opacity, "
                           "scaling and rotation")
    return _C_ext.rasterize_gaussians_raw(background, xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                                          scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                          image_width, degree, campos, debug, binning_capacity)


def rasterize_gaussians_raw_backward(background, xyz, radii, features_dc, features_rest, opacity_raw, scaling_raw,
                                     rotation_raw, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                     dL_dout_color, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, *,
                                     binning_capacity=0, accumulate_into=None, no_zero_fill=False):
    """Gradients w.r.t. the stored tensors: (means2D, xyz, features_dc, features_rest, opacity, scaling, rotation).
    accumulate_into: {"means2D","xyz","opacity","scaling","rotation": tensor, "features": (dc_grad, rest_grad)} adds
    in place (slot returned as None)."""
    acc = _NONE8
    if accumulate_into:
        f = accumulate_into.get("features") or (None, None)
        acc = [accumulate_into.get("means2D"), accumulate_into.get("xyz"), f[0], f[1], accumulate_into.get("opacity"),
               accumulate_into.get("scaling"), accumulate_into.get("rotation")]
    return tuple(_C_ext.rasterize_gaussians_raw_backward(
        background, xyz, radii, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw, scale_modifier, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, dL_dout_color, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
        binning_capacity, acc, bool(no_zero_fill)))


# one Adam step whose gradients are valid only in the rows of the Gaussians one view visited (lr_adam_step_masked)
adam_step_masked = _C_ext.adam_step_masked
mark_visible = _C_ext.mark_visible
check = _C_ext.check
# the operator with its autograd node compiled (csrc/torch_ext.cpp RasterizeFn); returns (color, radii, depth, geom), the
# call's num_rendered is read with last_num_rendered()
rasterize_autograd = _C_ext.rasterize_autograd
# forward + backward of one view in one call for a caller that holds dL/dcolor up front; [] = an input does not qualify
rasterize_view_step = _C_ext.rasterize_view_step
last_num_rendered = _C_ext.last_num_rendered


def request_early_header():
    """The next async-mode forward on this thread posts a header ticket right after its compaction scan (lr_request_early_header)."""
    _C_ext.request_early_header()


def take_early_ticket():
    """Ticket of that early header copy, or -1."""
    return _C_ext.take_early_ticket()


def last_forward_ticket():
    """Ticket (for header_poll) of the last async-mode forward on this thread, from the library's forward log: costs nothing --
    the scan kernel itself leaves the header in host-visible memory (lr_forward_ticket); -1 if there was none."""
    return _C_ext.last_forward_ticket()


def header_post(geomBuffer):
    """Non-blocking read-back of a forward's header on the current stream (lr_header_post); returns a ticket."""
    return _C_ext.header_post(geomBuffer)


def header_poll(ticket, block=False):
    """None while the read-back is in flight (block=False), else the 8 header words (the ticket is released)."""
    return _C_ext.header_poll(ticket, block)
