"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/lucid_raster.h declares; the Python operator mirrors the reference's API surface and error
behaviour; nothing in the product path falls back to the CPU or touches the oracle."""
import ast
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from luciddreamer_amd import build
    return build.build()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "lucid_raster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|size_t|char\s*\*|const char\s*\*)\s+(lr_[a-z0-9_]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_reference_entry_points():
    names = _declared_functions()
    for must in ("lr_forward", "lr_backward", "lr_mark_visible", "lr_dist2", "lr_geom_bytes", "lr_img_bytes",
                 "lr_binning_bytes", "lr_check", "lr_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    for name in _declared_functions():
        assert hasattr(L, name), f"{name} declared in include/lucid_raster.h but not exported"
    from luciddreamer_amd import _lib
    for name in _lib.EXPORTS:
        assert hasattr(L, name)


def test_size_queries_are_pure_host_functions(built_lib):
    from luciddreamer_amd import _lib
    L = _lib.lib()
    assert L.lr_version().decode().startswith("luciddreamer_amd")
    g1, g2 = L.lr_geom_bytes(1000), L.lr_geom_bytes(2000)
    assert 0 < g1 < g2 and g1 % 256 == 0
    assert L.lr_img_bytes(1920, 1080) >= 1920 * 1080 * 8
    assert L.lr_binning_bytes(10) < L.lr_binning_bytes(10_000_000)
    # per tile instance: 16 B sort ping-pong + 4 B Gaussian id + 48 B gradient slot (reference: ~24 B + sort temp,
    # and 9 global atomics per pixel pair instead of the slot)
    # 64 B per instance (list, sort words / quadrant tests, Gaussian id, 48-byte gradient slot) + 16 B: one 4 KB checkpoint and
    # one list entry per 256 instances for the segments of long lists (common.h BWD_SEG)
    assert (L.lr_binning_bytes(10_000_000) - L.lr_binning_bytes(0)) // 10_000_000 <= 84


def test_settings_tuple_matches_reference_fields():
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def _settings():
    from depth_diff_gaussian_rasterization_min import GaussianRasterizationSettings
    eye = torch.eye(4)
    return GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, eye, eye, 0, torch.zeros(3), False, False)


def test_argument_validation_mirrors_reference():
    from depth_diff_gaussian_rasterization_min import GaussianRasterizer
    r = GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    o = torch.zeros(4, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=o, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=o, shs=torch.zeros(4, 16, 3), colors_precomp=m, scales=m,
          rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m, rotations=torch.zeros(4, 4),
          cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m)


def test_cpu_tensors_are_rejected_not_silently_computed():
    """No CPU/PyTorch fallback: host tensors must fail loudly."""
    from depth_diff_gaussian_rasterization_min import GaussianRasterizer
    r = GaussianRasterizer(_settings())
    m = torch.rand(4, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        r(means3D=m, means2D=m, opacities=torch.rand(4, 1), colors_precomp=m, scales=m, rotations=torch.rand(4, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        from luciddreamer_amd import _C
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 2), m, m, m, m, 1.0, m, m, m, 0.5, 0.5, 8, 8, m, 0, m,
                               False, False)
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="HIP device"):
        distCUDA2(m)


def test_compiled_binding_is_the_one_in_tree():
    """The per-view entry points go through the compiled module luciddreamer_amd/_C_ext*.so (csrc/torch_ext.cpp), which
    links the in-tree C-ABI library; luciddreamer_amd._C is only its keyword adapter."""
    from luciddreamer_amd import _C, _C_ext, _lib
    assert os.path.dirname(os.path.abspath(_C_ext.__file__)) == os.path.join(ROOT, "luciddreamer_amd")
    assert _C_ext.version() == _lib.lib().lr_version().decode()
    assert _C.mark_visible is _C_ext.mark_visible and _C.check is _C_ext.check
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "rasterize_gaussians_raw",
                 "rasterize_gaussians_raw_backward"):
        assert callable(getattr(_C_ext, name)) and callable(getattr(_C, name))


def test_missing_library_fails_loudly(monkeypatch):
    from luciddreamer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblucid_raster.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_two_builds_of_the_library_in_one_process_are_refused(tmp_path):
    """LR_LIB_DIR without the same directory on LD_LIBRARY_PATH: ctypes loads the named build, the compiled binding the default
    one -- two sets of library state in one process (and, measured, a slower step).  Refused at the second load, whichever it
    is; with both variables set (tools/diag_env.sh) the process is on one build."""
    import shutil
    import subprocess
    import sys
    other = tmp_path / "lib_other"
    other.mkdir()
    shutil.copy(os.path.join(ROOT, "luciddreamer_amd", "lib", "liblucid_raster.so"), other / "liblucid_raster.so")
    code = "from luciddreamer_amd import _lib; _lib.lib(); from luciddreamer_amd import _C; print('LOADED', _lib.LIB_PATH)"
    env = dict(os.environ, LR_LIB_DIR=str(other), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "two copies of liblucid_raster.so" in r.stderr, r.stdout[-500:] + r.stderr[-1500:]
    code2 = "from luciddreamer_amd import _C, _lib; _lib.lib(); print('LOADED', _lib.LIB_PATH)"      # the other order of the loads
    r = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "two copies of liblucid_raster.so" in r.stderr, r.stdout[-500:] + r.stderr[-1500:]
    env["LD_LIBRARY_PATH"] = str(other) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and f"LOADED {other}" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
    may import it."""
    offenders = []
    for pkg in ("luciddreamer_amd", "depth_diff_gaussian_rasterization_min", "simple_knn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for fn in files:
                if not fn.endswith(".py"):
                    continue
                tree = ast.parse(open(os.path.join(dirpath, fn)).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom) and node.module:
                        mods = [node.module]
                    if any(m == "oracle" or m.startswith("oracle.") for m in mods):
                        offenders.append(os.path.join(dirpath, fn))
    assert not offenders, offenders
    # bench.py: oracle only inside run_cpu_baseline
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
            assert not any(m.startswith("oracle") for m in mods)
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        if uses:
            assert fn.name == "run_cpu_baseline"


def test_algorithmic_byte_model_matches_baseline_md():
    """BASELINE.md section 3 worked value for C2: B_f = 171 MB, B_b = 165 MB."""
    import bench
    P, V, R, N, T, K, M = 100_000, 74_350, 1_110_000, 1920 * 1080, 8160, 16, 16
    b_f, b_b = bench.path_bytes(P, V, R, N, T, K, M)
    assert abs(b_f - 171e6) / 171e6 < 0.02 and abs(b_b - 165e6) / 165e6 < 0.02
    assert bench.stage_bytes("render_bwd", P, V, R, N, T, K, M) == 40 * R + 20 * N + 44 * V


def test_division_free_index_arithmetic_of_the_binning_is_exact():
    """The device code maps bit / pair j of a tile rectangle of width w to (row, column) without an integer division:
    tilebin.hip walk_chunk uses (j * (65536 / w + 1)) >> 16 for masked rectangles (j < 64, w <= 64, common.h HitRec) and
    preprocess.hip's pooled count (j * (32768 / w + 1)) >> 15 with the reciprocal packed into 16 bits (j < 96, w <= 96,
    CULL_MAX_TILES).  Exhaustive check of both, and of the constants they rest on."""
    text = open(os.path.join(ROOT, "luciddreamer_amd", "csrc", "common.h")).read()
    assert re.search(r"HIT_MASK_TILES\s*=\s*64\b", text) and re.search(r"CULL_MAX_TILES\s*=\s*96\b", text)
    for w in range(1, 65):
        r = 65536 // w + 1
        assert all((j * r) >> 16 == j // w for j in range(64)), w
    for w in range(1, 97):
        r = 32768 // w + 1
        assert r < 65536 and all((j * r) >> 15 == j // w for j in range(96)), w
    # bitonic comparator indices by shifts (tilebin.hip bitonic_sort) == the textbook division form
    for lk in range(1, 8):
        k = 1 << lk
        for lj in range(lk - 1, -1, -1):
            j = 1 << lj
            for c in range(64):
                hi, lo = c >> lj, c & (j - 1)
                if lj == lk - 1:
                    assert (hi << lk) + lo == (c // j) * k + (c % j)
                else:
                    assert (hi << (lj + 1)) + lo == (c // j) * (j << 1) + (c % j)


def test_bench_reads_rocprofv3_counter_files_and_picks_the_timed_kernel(tmp_path):
    """bench.py's live byte-counter leg: the csv rocprofv3 --pmc writes (one row per dispatch and counter) averaged per kernel and
    launch, and the record of the kernel a stage timing belongs to (the blend backward has several shapes: the roofline leg names
    the one the headline launched)."""
    import bench
    d = tmp_path / "FETCH_SIZE" / "box"
    d.mkdir(parents=True)
    head = "Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    rows = [(1, "void lr::(anonymous namespace)::k_render_bwd_tile(int, int) [clone .kd]", "FETCH_SIZE", 50000.0),
            (2, "void lr::(anonymous namespace)::k_render_bwd<false, true, false>(int, int)", "FETCH_SIZE", 30000.0),
            (3, "void lr::(anonymous namespace)::k_render_bwd<false, true, false>(int, int)", "FETCH_SIZE", 34000.0),
            (3, "void lr::(anonymous namespace)::k_render_bwd<false, true, false>(int, int)", "SQ_WAVES", 7.0),
            (4, "__amd_rocclr_copyBuffer", "FETCH_SIZE", 1.0)]
    (d / "123_FETCH_SIZE_counter_collection.csv").write_text(
        head + "".join(f'{i},{i},0,"{k}",{c},{v},0,10\n' for i, k, c, v in rows))
    acc = {}
    assert bench.read_counter_csv(str(tmp_path / "FETCH_SIZE"), "FETCH_SIZE", acc) == 4          # the SQ_WAVES row is not this pass's
    assert acc["k_render_bwd<false, true, false>"]["FETCH_SIZE"] == [64000.0, 2] and acc["k_render_bwd_tile"]["FETCH_SIZE"] == [50000.0, 1]
    mean = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
    # the kernel a stage timing belongs to: the shape the timed leg launched when the caller names it (the multi-stream
    # headline at 1080p launches k_render_bwd_tile), else k_<stage><...> first
    assert bench.pick_kernel(mean, "render_bwd") == ("k_render_bwd<false, true, false>", {"FETCH_SIZE": 32000.0})
    assert bench.pick_kernel(mean, "render_bwd", "k_render_bwd_tile") == ("k_render_bwd_tile", {"FETCH_SIZE": 50000.0})
    assert bench.pick_kernel(mean, "gauss_bwd") == (None, {})
    assert bench.read_counter_csv(str(tmp_path / "nothing_here"), "FETCH_SIZE", {}) == 0
