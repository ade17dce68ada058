"""ctypes front-end of the CPU oracle (oracle/raster_oracle.c, oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product packages (luciddreamer_amd/,
depth_diff_gaussian_rasterization_min/, simple_knn/).  Parity of the restatement is pinned by
tests/test_oracle_ref.py against oracle/_ref (the reference's own .cu sources compiled for the
host, oracle/build_ref.py) and against the fixtures under tests/golden/.

Two backends share this front-end: "port" = liboracle.so (the restatement, default) and
"ref" = oracle/_ref/libref_raster.so (see oracle/ref.py); both export the same C signatures.

The argument lists mirror the reference's CudaRasterizer::Rasterizer::{forward,backward,
markVisible} (RAST/cuda_rasterizer/rasterizer.h:24-86) with numpy float32 arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile liboracle.so with gcc (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "knn_oracle.c", "Makefile")]
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB_PATH


STATE_ARRAYS = ("depths", "clamped", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched",
                "point_list", "point_list_keys", "ranges", "final_T", "n_contrib")


class Backend:
    """One shared library exporting <prefix>forward / backward / mark_visible / dist2 / state_* accessors."""

    def __init__(self, path, prefix, has_fragile):
        L = ctypes.CDLL(path)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        self.prefix, self.has_fragile, self.L = prefix, has_fragile, L

        def sym(name, restype, argtypes):
            f = getattr(L, prefix + name)
            f.restype, f.argtypes = restype, argtypes
            setattr(self, name, f)
        sym("forward", ci, [ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp,
                            cf, cf, ci, vp, vp, vp, ctypes.POINTER(vp)])
        sym("backward", None, [vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp,
                               cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp])
        sym("mark_visible", None, [ci, vp, vp, vp, vp])
        sym("state_free", None, [vp])
        sym("state_R", ci, [vp])
        for name in STATE_ARRAYS + (("fragile",) if has_fragile else ()):
            sym("state_" + name, vp, [vp])
        sym("dist2", None, [ci, vp, vp])


def lib():
    """The restatement ("port") backend."""
    global _lib
    if _lib is None:
        build()
        _lib = Backend(_LIB_PATH, "oracle_", True)
    return _lib


def set_accum_f32(on):
    """True: the backward sums its per-pixel terms in float32 in the order of a one-block-at-a-time run of the
    reference (bit-comparable with oracle/_ref on one thread); False (default): double accumulators."""
    lib().L.oracle_set_accum_f32(int(bool(on)))


def _f32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)


class ForwardResult:
    """Outputs + the opaque state the backward needs (freed on garbage collection)."""

    def __init__(self, backend):
        self._state = None
        self._backend = backend

    def __del__(self):
        st, self._state = getattr(self, "_state", None), None
        if st and _lib is not None:          # _lib is None again during interpreter shutdown
            try:
                self._backend.state_free(st)
            except Exception:
                pass

    def _arr(self, name, dtype, count):
        p = getattr(self._backend, "state_" + name)(self._state)
        if count == 0:
            return np.zeros((0,), dtype=dtype)
        buf = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=count).copy()

    # stage-level views (copies)
    def stage(self):
        P, N, R = self.P, self.W * self.H, self.num_rendered
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        return dict(
            depths=self._arr("depths", np.float32, P),
            clamped=self._arr("clamped", np.uint8, 3 * P).reshape(P, 3),
            means2D=self._arr("means2D", np.float32, 2 * P).reshape(P, 2),
            cov3D=self._arr("cov3D", np.float32, 6 * P).reshape(P, 6),
            conic_opacity=self._arr("conic_opacity", np.float32, 4 * P).reshape(P, 4),
            rgb=self._arr("rgb", np.float32, 3 * P).reshape(P, 3),
            tiles_touched=self._arr("tiles_touched", np.uint32, P),
            point_list=self._arr("point_list", np.uint32, max(R, 0)),
            point_list_keys=self._arr("point_list_keys", np.uint64, max(R, 0)),
            ranges=self._arr("ranges", np.uint32, 2 * T).reshape(T, 2),
            final_T=self._arr("final_T", np.float32, N).reshape(self.H, self.W),
            n_contrib=self._arr("n_contrib", np.uint32, N).reshape(self.H, self.W),
            fragile=(self._arr("fragile", np.uint8, N) if self._backend.has_fragile
                     else np.zeros(N, np.uint8)).reshape(self.H, self.W),
        )


def forward(background, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
            cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
            sh, degree, campos, prefiltered=False, backend=None):
    """Same argument order as _C.rasterize_gaussians (RAST/rasterize_points.h:18-38), numpy in/out."""
    L = backend or lib()
    means3D = _f32(means3D)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    bg, col, op = _f32(background), _f32(colors_precomp), _f32(opacities)
    sc, rot, cov = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    view, proj, cam, shs = _f32(viewmatrix), _f32(projmatrix), _f32(campos), _f32(sh)
    M = 0
    if shs is not None and shs.size != 0:
        M = shs.shape[1]
    out_color = np.zeros((3, H, W), np.float32)
    out_depth = np.zeros((1, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    res = ForwardResult(L)
    res.P, res.W, res.H = P, W, H
    res.num_rendered = 0
    if P != 0:
        st = ctypes.c_void_p()
        R = L.forward(P, int(degree), M, _ptr(bg), W, H, _ptr(means3D), _ptr(shs), _ptr(col),
                             _ptr(op), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(cov),
                             _ptr(view), _ptr(proj), _ptr(cam), float(tan_fovx), float(tan_fovy),
                             int(bool(prefiltered)), _ptr(out_color), _ptr(out_depth), _ptr(radii),
                             ctypes.byref(st))
        res._state = st
        if R == -2:
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
        res.num_rendered = R
    res.color, res.depth, res.radii = out_color, out_depth, radii
    # keep inputs alive / handy for backward
    res._inputs = dict(bg=bg, means3D=means3D, col=col, sc=sc, rot=rot, cov=cov, view=view, proj=proj,
                       cam=cam, shs=shs, M=M, degree=int(degree), scale_modifier=float(scale_modifier),
                       tan_fovx=float(tan_fovx), tan_fovy=float(tan_fovy))
    return res


def blend_f64(res, background, colors_precomp=None):
    """The blend of a finished forward (restatement backend) evaluated in float64 on its float32 records and tile lists
    (oracle_blend_f64): (3, H, W) float64.  A yardstick for float32 evaluations that disagree, not a reference output."""
    L = lib()
    if res._backend is not L:
        raise ValueError("blend_f64 needs a ForwardResult of the restatement backend")
    f = L.L.oracle_blend_f64
    f.restype, f.argtypes = None, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p]
    out = np.zeros((3, res.H, res.W), np.float64)
    bg, col = _f32(background), _f32(colors_precomp)
    if res._state:
        f(res._state, res.W, res.H, _ptr(bg), _ptr(col), out.ctypes.data_as(ctypes.c_void_p))
    else:
        out[:] = bg.reshape(3, 1, 1)
    return out


def backward(res, dL_dout_color, dL_dout_depth=None):
    """Returns the reference's 8-tuple order (RAST/rasterize_points.cu:199):
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    plus dL_dconic as a 9th element for stage tests."""
    L = res._backend
    i = res._inputs
    P, H, W, M = res.P, res.H, res.W, i["M"]
    g = _f32(dL_dout_color)
    gd = _f32(dL_dout_depth) if dL_dout_depth is not None else np.zeros((1, H, W), np.float32)
    dmeans3D = np.zeros((P, 3), np.float32)
    dmeans2D = np.zeros((P, 3), np.float32)
    dcolors = np.zeros((P, 3), np.float32)
    dconic = np.zeros((P, 2, 2), np.float32)
    dopacity = np.zeros((P, 1), np.float32)
    dcov3D = np.zeros((P, 6), np.float32)
    dsh = np.zeros((P, M, 3), np.float32)
    dscales = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32)
    if P != 0:
        L.backward(res._state, P, i["degree"], M, _ptr(i["bg"]), W, H, _ptr(i["means3D"]),
                          _ptr(i["shs"]), _ptr(i["col"]), _ptr(i["sc"]), i["scale_modifier"], _ptr(i["rot"]),
                          _ptr(i["cov"]), _ptr(i["view"]), _ptr(i["proj"]), _ptr(i["cam"]),
                          i["tan_fovx"], i["tan_fovy"], _ptr(g), _ptr(gd),
                          _ptr(dmeans2D), _ptr(dconic), _ptr(dopacity), _ptr(dcolors), _ptr(dmeans3D),
                          _ptr(dcov3D), _ptr(dsh) if M else None, _ptr(dscales), _ptr(drot))
    return dmeans2D, dcolors, dopacity, dmeans3D, dcov3D, dsh, dscales, drot, dconic


def mark_visible(means3D, viewmatrix, projmatrix, backend=None):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros((P,), np.uint8)
    if P:
        (backend or lib()).mark_visible(P, _ptr(means3D), _ptr(_f32(viewmatrix)), _ptr(_f32(projmatrix)), _ptr(out))
    return out.astype(bool)


def dist2(points, backend=None):
    """simple_knn distCUDA2 contract (KNN/spatial.cu:15-26)."""
    pts = _f32(points)
    P = pts.shape[0]
    out = np.zeros((P,), np.float32)
    if P:
        (backend or lib()).dist2(P, _ptr(pts), _ptr(out))
    return out
