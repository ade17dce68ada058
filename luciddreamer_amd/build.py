"""In-tree hipcc build of the C-ABI library luciddreamer_amd/lib/liblucid_raster.so (gfx950 only).

    python -m luciddreamer_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so stays in-tree (git-ignored) so it travels to the GPU
box with the repo snapshot; nothing is JIT-compiled at run time.
"""
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB_PATH = os.path.join(LIBDIR, "liblucid_raster.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-I", INCLUDE]
# per-file extra flags
SOURCES = {
    # bit-exact projection / conic / radius vs the CPU oracle: no FMA contraction in this TU
    "preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": [],          # the generic radix sort (simple-knn's Morton order)
    "tilebin.hip": [],          # compaction, tile partition, per-tile sort
    # scalar per-pixel state on purpose (see the kernel): keep the SLP vectoriser from re-packing it
    "render_fwd.hip": ["-fno-slp-vectorize"],
    "render_bwd.hip": ["-fno-slp-vectorize"],
    "gauss_bwd.hip": [],
    "knn.hip": [],
    # the separable window sums are long fma chains: packed f32 costs two issue slots plus the moves that form the pairs
    "loss.hip": ["-fno-slp-vectorize"],
    "rows.hip": [],
    "adam.hip": [],
    "api.hip": [],
}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """12 hex digits over every kernel / host source of the library: lr_version() carries it, so that a counter file
    (profiles/pmc_c3.json) or a bench line can be tied to the exact kernels it was measured on (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha1()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for path in [os.path.join(CSRC, f) for f in names] + [os.path.join(INCLUDE, "lucid_raster.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def _write_hash_header():
    path = os.path.join(OBJDIR, "lr_src_hash.h")
    text = f'#define LR_SRC_HASH "{source_hash()}"\n'
    old = open(path).read() if os.path.exists(path) else None
    if old != text:                       # untouched when the sources are: api.o is only rebuilt when the hash moves
        with open(path, "w") as f:
            f.write(text)
    return path


DIAG_LIBDIR = os.path.join(_HERE, "lib_diag")      # the diagnostics build (retired kernels, LR_* environment overrides): tools only


def build(force=False, verbose=False, diagnostics=False):
    """The product library lib/liblucid_raster.so, or with diagnostics=True the same sources with -DLR_DIAGNOSTICS into
    lib_diag/liblucid_raster.so (A/B partners of tools/ab_bench.py; never loaded unless LR_LIB_DIR says so, tools/diag_env.sh)."""
    libdir = DIAG_LIBDIR if diagnostics else LIBDIR
    objdir = os.path.join(libdir, "obj")
    lib_path = os.path.join(libdir, "liblucid_raster.so")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    hash_header = _write_hash_header()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "lucid_raster.h"), os.path.abspath(__file__)]
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        if diagnostics:
            extra = extra + ["-DLR_DIAGNOSTICS"]
        objs.append(op)
        deps = [sp] + headers + ([hash_header] if src == "api.hip" else [])
        if force or _stale(op, deps):
            cmd = [cc, "-c", sp, "-o", op] + COMMON_FLAGS + ["-I", OBJDIR] + extra + os.environ.get("LR_EXTRA_HIPCC_FLAGS", "").split()
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s} ---\n{o}" for s, o in failed))
    if force or procs or _stale(lib_path, objs):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib_path


EXT_SRC = os.path.join(CSRC, "torch_ext.cpp")


def ext_path():
    import sysconfig
    return os.path.join(_HERE, "_C_ext" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ext(force=False, verbose=False):
    """In-tree build of luciddreamer_amd/_C_ext*.so: the compiled torch binding over the C-ABI (csrc/torch_ext.cpp;
    host code only, g++ against the torch / HIP headers, linked to lib/liblucid_raster.so with an $ORIGIN rpath)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    out = ext_path()
    deps = [EXT_SRC, os.path.join(INCLUDE, "lucid_raster.h"), os.path.abspath(__file__)]
    if not force and not _stale(out, deps):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", EXT_SRC, "-o", out,
           "-DTORCH_EXTENSION_NAME=_C_ext", "-DTORCH_API_INCLUDE_EXTENSION_H", "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I", INCLUDE, "-I", os.path.join(rocm, "include"),
           "-I", sysconfig.get_paths()["include"]]
    for inc in ce.include_paths():
        cmd += ["-isystem", inc]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd += ["-L", torch_lib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-L", LIBDIR, "-llucid_raster", "-Wl,-rpath,$ORIGIN/lib", f"-Wl,-rpath,{torch_lib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--diagnostics" in sys.argv:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, diagnostics=True))
        sys.exit(0)
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
    print(build_ext(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
