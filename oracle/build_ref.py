#!/usr/bin/env python
"""Recipe for oracle/_ref: compile the reference's OWN sources, where they lie under /root/reference, for the host.

TEST INFRASTRUCTURE ONLY.  Output: oracle/_ref/libref_raster.so (git-ignored, travels to the GPU box like our own .so
files).  No reference source text is written anywhere: each .cu file is streamed from /root/reference through one
rewrite and into g++'s standard input.

What is compiled (all unchanged reference text):
    RAST/cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu   (+ their headers auxiliary.h, config.h, ...)
    KNN/simple_knn.cu                                                   (+ simple_knn.h)
against oracle/ref_shim/ (our stand-ins for cuda_runtime.h, cooperative_groups.h, cub/cub.cuh, thrust, glm/glm.hpp; see
cuda_host_shim.h and glm/glm.hpp there), plus oracle/ref_shim/ref_capi.cpp, the extern "C" face.

The one rewrite: CUDA's launch syntax is not C++, so `kernel<T> << <grid, block >> > (args)` becomes
`shim::launch(kernel<T>, grid, block)(args)` -- the same kernel, the same launch configuration, the same arguments.
Flags: -O2 -ffp-contract=off (no FMA contraction: float32 in source operand order; nvcc would contract).

If /root/reference is absent (the GPU box) nothing is built and the prebuilt library, if present, is used as is.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LUCID_REFERENCE_ROOT", "/root/reference")
RAST = os.path.join(REF, "submodules", "depth-diff-gaussian-rasterization-min")
KNN = os.path.join(REF, "submodules", "simple-knn")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libref_raster.so")
SHIM = os.path.join(HERE, "ref_shim")

SOURCES = [
    os.path.join(RAST, "cuda_rasterizer", "forward.cu"),
    os.path.join(RAST, "cuda_rasterizer", "backward.cu"),
    os.path.join(RAST, "cuda_rasterizer", "rasterizer_impl.cu"),
    os.path.join(KNN, "simple_knn.cu"),
]
LAUNCH = re.compile(r"(\b\w+(?:<[^<>]*>)?)\s*<<\s*<(.*?)>>\s*>\s*\(")
CXXFLAGS = ["-std=c++17", "-O2", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-w",
            "-I", SHIM, "-iquote", os.path.join(RAST, "cuda_rasterizer"), "-iquote", KNN]


def available():
    return all(os.path.exists(s) for s in SOURCES)


def _stale():
    if not os.path.exists(LIB):
        return True
    deps = list(SOURCES) + [os.path.abspath(__file__)]
    for root, _, files in os.walk(SHIM):
        deps += [os.path.join(root, f) for f in files]
    for d in (os.path.join(RAST, "cuda_rasterizer"), KNN):
        deps += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    """Returns the library path, or None when neither the reference nor a prebuilt library is there."""
    if not available():
        return LIB if os.path.exists(LIB) else None
    if not force and not _stale():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        text = open(src, encoding="utf-8", errors="replace").read()
        text, n = LAUNCH.subn(r"shim::launch(\1, \2)(", text)
        assert n > 0 or src.endswith("forward.h"), src
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        cmd = ["g++", "-x", "c++", "-c", "-o", obj] + CXXFLAGS + ["-"]
        if verbose:
            print(" ".join(cmd), f"   # {src}: {n} launches rewritten")
        subprocess.run(cmd, input=f'#line 1 "{src}"\n{text}'.encode(), check=True)
        objs.append(obj)
    capi = os.path.join(OUT_DIR, "ref_capi.o")
    subprocess.run(["g++", "-x", "c++", "-c", "-o", capi] + CXXFLAGS + [os.path.join(SHIM, "ref_capi.cpp")], check=True)
    subprocess.run(["g++", "-shared", "-fopenmp", "-o", LIB] + objs + [capi, "-lm"], check=True)
    for o in objs + [capi]:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path if path else "reference sources not found and no prebuilt oracle/_ref/libref_raster.so")
