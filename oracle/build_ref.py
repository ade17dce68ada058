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


SHIM_HIP = os.path.join(HERE, "ref_shim_hip")
LIB_DEV = os.path.join(OUT_DIR, "libref_raster_gfx950.so")
LIB_DEV_FMA = os.path.join(OUT_DIR, "libref_raster_gfx950_fma.so")


def build_device(force=False, verbose=False, contract="off"):
    """The same reference sources compiled by hipcc for gfx950 (streamed through ONE whitespace rewrite: the reference
    spells its launches `kernel << <grid, block >> > (...)`, which nvcc accepts and clang does not -> `kernel<<<grid, block>>>(`) against
    oracle/ref_shim_hip/ (CUDA header names -> ROCm headers, cub -> hipCUB, the GLM subset with device qualifiers) plus
    ref_capi_hip.cpp.  Output: oracle/_ref/libref_raster_gfx950.so -- the reference's own kernels on the MI355X, used as a
    device-side checker at full sizes and as the `reference_on_device` baseline of bench.py.  Returns the path or None.

    contract="fast": a second library, libref_raster_gfx950_fma.so, with the compiler's DEFAULT floating-point contraction
    (hipcc -ffp-contract=fast; nvcc's default -fmad=true is the same choice, i.e. how the reference's own setup.py builds
    it).  Only used to measure how far the reference moves under its own build flags (tests/test_gpu_ref_selfcal.py)."""
    LIB_DEV = LIB_DEV_FMA if contract == "fast" else globals()["LIB_DEV"]
    if not available():
        return LIB_DEV if os.path.exists(LIB_DEV) else None
    deps = list(SOURCES) + [os.path.abspath(__file__)]
    for root in (SHIM_HIP, os.path.join(SHIM, "glm")):
        for r, _, files in os.walk(root):
            deps += [os.path.join(r, f) for f in files]
    if not force and os.path.exists(LIB_DEV) and all(os.path.getmtime(d) <= os.path.getmtime(LIB_DEV) for d in deps):
        return LIB_DEV
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=off like the host build: float32 in source operand order (with contraction the per-Gaussian records
    # differ in their last bits and ~0.1 % of the pixels of a 1080p view move by more than 1e-5)
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=" + contract, "-std=c++17", "-fPIC", "-w", "-I", SHIM_HIP,
             "-I", os.path.join(RAST, "cuda_rasterizer"), "-I", KNN]
    objs = []
    procs = []
    # clang's HIP driver reads its input twice (host and device pass), so the rewritten text cannot come through a pipe:
    # it goes to a scratch file under the git-ignored oracle/_ref/ that is deleted as soon as the object exists
    tmp = []
    try:
        for src in SOURCES + [os.path.join(SHIM_HIP, "ref_capi_hip.cpp")]:
            obj = os.path.join(OUT_DIR, f"dev_{contract}_" + os.path.basename(src) + ".o")
            text = open(src, encoding="utf-8", errors="replace").read()
            text = LAUNCH.sub(r"\1<<<\2>>>(", text)
            scratch = os.path.join(OUT_DIR, f"scratch_{contract}_" + os.path.basename(src) + ".hip")
            with open(scratch, "w") as f:
                f.write(f'#line 1 "{src}"\n{text}')
            tmp.append(scratch)
            cmd = [hipcc, "-x", "hip", "-c", scratch, "-o", obj] + flags + ["-iquote", os.path.dirname(src)]
            if verbose:
                print(" ".join(cmd), f"  # = {src}", flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            objs.append(obj)
        for src, pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{out[-6000:]}")
    finally:
        for t in tmp:
            if os.path.exists(t):
                os.remove(t)
    subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_DEV] + objs, check=True)
    for o in objs:
        os.remove(o)
    return LIB_DEV


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path if path else "reference sources not found and no prebuilt oracle/_ref/libref_raster.so")
    print(build_device(force="--force" in sys.argv, verbose=True))
