"""The reference's own PYTHON layer as a checker: render() (R/gaussian_renderer/__init__.py:18-104), GaussianModel
(R/scene/gaussian_model.py), the autograd op (RAST/depth_diff_gaussian_rasterization_min/__init__.py) and the loss
(R/utils/loss.py), imported UNCHANGED from /root/reference -- or, on the GPU box where /root/reference does not exist,
from the copies `stage()` placed under oracle/_ref/py/ (git-ignored build output, made in the build container by
__graft_entry__.build(); it travels with the snapshot like the built libraries).

TEST INFRASTRUCTURE ONLY: used by tests/ and tests/golden/*.py, never by the product packages.

Two ways to run it:
  * on CPU tensors, with the rasterizer extension `_C` replaced by a stand-in over a CPU backend of oracle/oracle.py --
    "ref" (oracle/_ref: the reference's .cu sources compiled for the host) or "port" (the restatement).  The reference
    hard-codes device="cuda" (gaussian_model.py, gaussian_renderer/__init__.py:26, utils/general.py); `cuda_as_cpu()`
    maps that to the CPU for the duration of a with-block.
  * on the GPU box against THIS repository's packages (rasterizer="ours"): the zero-change route of INTEGRATION.md.
"""
import contextlib
import importlib
import importlib.util
import os
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LUCID_REFERENCE_ROOT", "/root/reference")
STAGED = os.path.join(HERE, "_ref", "py")
RAST_PY = os.path.join("submodules", "depth-diff-gaussian-rasterization-min", "depth_diff_gaussian_rasterization_min",
                       "__init__.py")
FILES = ["gaussian_renderer/__init__.py", "scene/gaussian_model.py", "utils/general.py", "utils/system.py", "utils/sh.py",
         "utils/graphics.py", "utils/loss.py", "arguments.py", RAST_PY]


def stage():
    """Build-container step: copy the handful of reference .py files the GPU-box tests import into oracle/_ref/py/."""
    if not os.path.isdir(REF):
        return STAGED if os.path.isdir(STAGED) else None
    for rel in FILES:
        dst = os.path.join(STAGED, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
    return STAGED


def root():
    if os.path.isdir(REF):
        return REF
    return STAGED if os.path.isfile(os.path.join(STAGED, FILES[0])) else None


def available():
    return root() is not None


@contextlib.contextmanager
def cuda_as_cpu():
    """Inside the block, device="cuda" / .cuda() mean the CPU (for running the reference's Python without a device)."""
    names = ["zeros", "ones", "empty", "full", "tensor", "as_tensor", "zeros_like", "ones_like", "normal", "rand",
             "randn", "arange", "eye"]
    saved = {n: getattr(torch, n) for n in names}
    saved_cuda, saved_mod_cuda, saved_cache = torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.empty_cache

    def wrap(f):
        def g(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return f(*a, **k)
        return g
    try:
        for n in names:
            setattr(torch, n, wrap(saved[n]))
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.empty_cache = lambda: None
        yield
    finally:
        for n in names:
            setattr(torch, n, saved[n])
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.empty_cache = saved_cuda, saved_mod_cuda, saved_cache


class CpuRasterizerExtension:
    """Stand-in for the compiled `_C` module (RAST/ext.cpp:15-19) on CPU tensors over a backend of oracle/oracle.py.
    Same three functions, same positional arguments, same return tuples (RAST/rasterize_points.h:18-68)."""

    def __init__(self, backend):
        from . import oracle, ref
        self._forward = ref.forward if backend == "ref" else oracle.forward
        self._backward = oracle.backward
        self._mark = ref.mark_visible if backend == "ref" else oracle.mark_visible
        self._live = {}
        self._next = 1

    @staticmethod
    def _np(t):
        return None if t is None or t.numel() == 0 else t.detach().contiguous().numpy()

    def rasterize_gaussians(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug):
        n = self._np
        res = self._forward(n(bg), n(means3D), n(colors), n(opacity), n(scales), n(rotations), scale_modifier,
                            n(cov3D_precomp), n(viewmatrix), n(projmatrix), tan_fovx, tan_fovy, image_height,
                            image_width, n(sh), degree, n(campos), prefiltered)
        handle = self._next
        self._next += 1
        self._live[handle] = res
        for old in [h for h in self._live if h <= handle - 4]:      # render-only callers never run a backward
            del self._live[old]
        tag = torch.tensor([handle], dtype=torch.int64)
        return (res.num_rendered, torch.from_numpy(res.color), torch.from_numpy(res.depth), torch.from_numpy(res.radii),
                tag, tag.clone(), tag.clone())

    def rasterize_gaussians_backward(self, bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree,
                                     campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
        res = self._live.pop(int(geomBuffer[0]))
        g = self._backward(res, self._np(dL_dout_color), self._np(dL_dout_depth))
        return tuple(torch.from_numpy(a) for a in g[:8])

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        return torch.from_numpy(self._mark(self._np(means3D), self._np(viewmatrix), self._np(projmatrix)))


class DeviceRasterizerExtension:
    """Stand-in for `_C` over the reference's OWN kernels on the GPU (oracle/ref_device.py: oracle/_ref compiled by hipcc
    for gfx950): the reference's Python op, render() and GaussianModel then run end to end on device tensors with nothing of
    this repository's rasterizer involved -- the device-side checker of the loss-curve tests."""

    def __init__(self):
        from . import ref_device
        self._rd = ref_device
        self._pool, self._live, self._next = [], {}, 1

    @staticmethod
    def _c(t):
        return None if t is None or t.numel() == 0 else t.detach().contiguous()

    def rasterize_gaussians(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug):
        c = self._c
        r = self._pool.pop() if self._pool else self._rd.Renderer()
        R, color, depth, radii = r.forward(c(bg), c(means3D), c(colors), c(opacity), c(scales), c(rotations), scale_modifier,
                                           c(cov3D_precomp), c(viewmatrix), c(projmatrix), tan_fovx, tan_fovy, image_height,
                                           image_width, c(sh), degree, c(campos), prefiltered)
        handle = self._next
        self._next += 1
        self._live[handle] = r
        for old in [h for h in self._live if h <= handle - 4]:
            self._pool.append(self._live.pop(old))
        tag = torch.tensor([handle], dtype=torch.int64)
        return R, color, depth, radii, tag, tag.clone(), tag.clone()

    def rasterize_gaussians_backward(self, bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree,
                                     campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
        r = self._live.pop(int(geomBuffer[0]))
        g = r.backward(dL_dout_color.contiguous())
        out = tuple(t.clone() for t in g[:8])
        self._pool.append(r)
        return out

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        raise NotImplementedError("markVisible is not wired for the device-side reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_OURS = ("utils", "scene", "gaussian_renderer", "arguments", "plyfile", "cv2", "depth_diff_gaussian_rasterization_min",
         "simple_knn")


@contextlib.contextmanager
def reference_modules(rasterizer="ours"):
    """Yields a namespace with the reference's modules imported unchanged: .gaussian_renderer (render), .gaussian_model
    (GaussianModel), .arguments (GSParams), .loss (l1_loss, ssim), .general, .sh, .rasterizer (the package render() uses).

    rasterizer="ours": `depth_diff_gaussian_rasterization_min` / `simple_knn` resolve to this repository (device run).
    rasterizer="ref" | "port": the reference's own Python op over a CPU stand-in for `_C`; simple_knn._C.distCUDA2 is
    served by the same backend.  Enter `cuda_as_cpu()` as well when no device is involved.
    rasterizer="refdev": the reference's own Python op over its own kernels on the GPU (oracle/ref_device.py)."""
    base = root()
    if base is None:
        raise RuntimeError("neither /root/reference nor oracle/_ref/py is present")
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _OURS}
    for k in list(saved):
        top = k.split(".")[0]
        if top in ("utils", "scene", "gaussian_renderer", "arguments") or \
                (rasterizer != "ours" and top in ("depth_diff_gaussian_rasterization_min", "simple_knn")):
            del sys.modules[k]
    try:
        if "plyfile" not in sys.modules:
            sys.modules["plyfile"] = _stub("plyfile", PlyData=object, PlyElement=object)
        if "cv2" not in sys.modules:
            sys.modules["cv2"] = _stub("cv2")
        scene = types.ModuleType("scene")                 # scene/__init__.py pulls in dataset readers; skip it
        scene.__path__ = [os.path.join(base, "scene")]
        sys.modules["scene"] = scene
        if rasterizer != "ours":
            from . import oracle, ref
            if rasterizer == "refdev":
                from . import ref_device
                ext = DeviceRasterizerExtension()
            else:
                ext = CpuRasterizerExtension(rasterizer)
            spec = importlib.util.spec_from_file_location(
                "depth_diff_gaussian_rasterization_min", os.path.join(base, RAST_PY),
                submodule_search_locations=[os.path.dirname(os.path.join(base, RAST_PY))])
            pkg = importlib.util.module_from_spec(spec)
            sys.modules["depth_diff_gaussian_rasterization_min"] = pkg
            sys.modules["depth_diff_gaussian_rasterization_min._C"] = ext
            spec.loader.exec_module(pkg)                  # runs `from . import _C` -> the stand-in
            if rasterizer == "refdev":
                dist_fn = lambda pts: ref_device.dist2(pts.detach())
            else:
                d2 = ref.dist2 if rasterizer == "ref" else oracle.dist2
                dist_fn = lambda pts: torch.from_numpy(d2(pts.detach().cpu().numpy()))
            knn = _stub("simple_knn")
            knn.__path__ = []
            knn._C = _stub("simple_knn._C", distCUDA2=dist_fn)
            sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn._C
        sys.path.insert(0, base)
        ns = types.SimpleNamespace()
        with (cuda_as_cpu() if not torch.cuda.is_available() else contextlib.nullcontext()):
            ns.general = importlib.import_module("utils.general")
            ns.sh = importlib.import_module("utils.sh")
            ns.loss = importlib.import_module("utils.loss")           # moves a 3x3 conv to "cuda" at import (:81-88)
            ns.arguments = importlib.import_module("arguments")
            ns.gaussian_model = importlib.import_module("scene.gaussian_model")
            ns.gaussian_renderer = importlib.import_module("gaussian_renderer")
        ns.rasterizer = sys.modules["depth_diff_gaussian_rasterization_min"]
        yield ns
    finally:
        if base in sys.path:
            sys.path.remove(base)
        for k in list(sys.modules):
            if k.split(".")[0] in _OURS:
                del sys.modules[k]
        sys.modules.update(saved)


class PointCloud:
    """What GaussianModel.create_from_pcd reads (R/utils/graphics.py BasicPointCloud: points, colors, normals)."""

    def __init__(self, points, colors):
        self.points, self.colors, self.normals = np.asarray(points), np.asarray(colors), np.zeros_like(points)
