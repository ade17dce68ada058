"""ctypes binding of liblucid_raster.so (the C-ABI declared in include/lucid_raster.h).

This is the reference-side binding a maintainer would add in place of the pybind11 module
`depth_diff_gaussian_rasterization_min._C` (RAST/ext.cpp:15-19).  There is NO CPU fallback: if the
HIP library is missing this module raises, and tensors that are not on a HIP device are rejected.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# one library instance per process: the compiled binding (_C_ext) is linked against this very file ($ORIGIN/lib).
# LR_LIB_DIR (with LD_LIBRARY_PATH pointing at the same directory, so that _C_ext resolves to the same file) switches the
# process to another build of it -- the diagnostics build of tools/ab_bench.py (tools/diag_env.sh sets both).
LIB_PATH = os.path.join(os.environ.get("LR_LIB_DIR") or os.path.join(_HERE, "lib"), "liblucid_raster.so")

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

LR_ERR_INVALID_ARG = -10
LR_ERR_HIP = -11
LR_ERR_PREFILTERED = -12
LR_ERR_OVERFLOW = -13
LR_ERR_ALLOC = -14
LR_NUM_RENDERED_ON_DEVICE = -1

_lib = None
_lock = threading.Lock()

EXPORTS = ("lr_last_error", "lr_version", "lr_geom_bytes", "lr_img_bytes", "lr_binning_bytes", "lr_forward",
           "lr_backward", "lr_forward_raw", "lr_backward_raw", "lr_mark_visible", "lr_check", "lr_dist2_workspace_bytes", "lr_dist2",
           "lr_profile_enable", "lr_profile_stage_name", "lr_profile_read", "lr_tune_set", "lr_last_launch_shapes", "lr_request_early_header",
           "lr_take_early_ticket", "lr_forward_ticket", "lr_backward_wait_event", "lr_step_begin", "lr_step_end", "lr_step_abort",
           "lr_views_workspace_bytes", "lr_views_accumulate", "lr_views_check",
           "lr_loss_workspace_bytes", "lr_l1_dssim_forward", "lr_l1_dssim_backward", "lr_l1_dssim_backward_weights",
           "lr_select_workspace_bytes", "lr_select_rows", "lr_pack_ply_rows", "lr_adam_step", "lr_adam_step_masked", "lr_densify_stats",
           "lr_views_train_workspace_bytes", "lr_views_train_accumulate", "lr_views_train_check")


def assert_single_copy():
    """One library instance per process: raises if two DIFFERENT liblucid_raster.so files are mapped -- LR_LIB_DIR set without the
    same directory on LD_LIBRARY_PATH leaves the compiled binding on the default build while ctypes loads the other one: two sets of
    per-stream scratch, forward logs and tuning state, wrong answers from every query that crosses them, and (measured, round 6)
    a C3 step 1.5 % slower.  Called behind both loads (here and in _C.py); whichever comes second sees both."""
    try:
        with open("/proc/self/maps") as f:
            paths = {ln.split()[-1] for ln in f if ln.rstrip().endswith("liblucid_raster.so")}
    except OSError:                                                          # no procfs: nothing to check with
        return
    real = {os.path.realpath(p) for p in paths}
    if len(real) > 1:
        raise RuntimeError("luciddreamer_amd: two copies of liblucid_raster.so in this process (" + ", ".join(sorted(real)) +
                           "): LR_LIB_DIR needs the same directory on LD_LIBRARY_PATH (tools/diag_env.sh sets both)")


def lib():
    """Load the library (built by `python -m luciddreamer_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"luciddreamer_amd: HIP library {LIB_PATH} is missing -- build it with "
                "`python -m luciddreamer_amd.build` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        assert_single_copy()
        vp, ci, cf, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
        L.lr_last_error.restype = ctypes.c_char_p
        L.lr_last_error.argtypes = []
        L.lr_version.restype = ctypes.c_char_p
        L.lr_version.argtypes = []
        L.lr_geom_bytes.restype = ctypes.c_size_t
        L.lr_geom_bytes.argtypes = [ci]
        L.lr_img_bytes.restype = ctypes.c_size_t
        L.lr_img_bytes.argtypes = [ci, ci]
        L.lr_binning_bytes.restype = ctypes.c_size_t
        L.lr_binning_bytes.argtypes = [ll]
        L.lr_forward.restype = ci
        L.lr_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp,      # allocators
                                 ci, ci, ci, vp, ci, ci,                         # P D M bg W H
                                 vp, vp, vp, vp, vp, cf, vp, vp,                 # means3D shs colors opac scales mod rot cov3D
                                 vp, vp, vp, cf, cf, ci,                         # view proj campos tanx tany prefiltered
                                 vp, vp, vp, ci, ll, vp]                         # out_color out_depth radii debug capacity stream
        L.lr_backward.restype = ci
        L.lr_backward.argtypes = [ci, ci, ci, ci, vp, ci, ci,                    # P D M R bg W H
                                  vp, vp, vp, vp, cf, vp, vp,                    # means3D shs colors scales mod rot cov3D
                                  vp, vp, vp, cf, cf, vp,                        # view proj campos tanx tany radii
                                  vp, vp, vp, vp, vp,                            # geom binning img dL_dpix dL_ddepth
                                  vp, vp, vp, vp, vp, vp, vp, vp, vp,            # 9 gradient outputs
                                  ci, ll, ctypes.c_uint, vp]                     # debug capacity accumulate_mask stream
        L.lr_forward_raw.restype = ci
        L.lr_forward_raw.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp,  # allocators
                                     ci, ci, ci, vp, ci, ci,                     # P D M bg W H
                                     vp, vp, vp, vp, vp, cf, vp,                 # xyz f_dc f_rest opacity scaling mod rotation
                                     vp, vp, vp, cf, cf,                         # view proj campos tanx tany
                                     vp, vp, vp, ci, ll, vp]                     # out_color out_depth radii debug capacity stream
        L.lr_backward_raw.restype = ci
        L.lr_backward_raw.argtypes = [ci, ci, ci, ci, vp, ci, ci,                # P D M R bg W H
                                      vp, vp, vp, vp, vp, cf, vp,                # xyz f_dc f_rest opacity scaling mod rotation
                                      vp, vp, vp, cf, cf, vp,                    # view proj campos tanx tany radii
                                      vp, vp, vp, vp,                            # geom binning img dL_dpix
                                      vp, vp, vp, vp, vp, vp, vp,                # 7 gradient outputs
                                      ci, ll, ctypes.c_uint, vp]                 # debug capacity accumulate_mask stream
        L.lr_mark_visible.restype = ci
        L.lr_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
        L.lr_check.restype = ci
        L.lr_check.argtypes = [vp, ctypes.POINTER(ll), vp]
        L.lr_dist2_workspace_bytes.restype = ctypes.c_size_t
        L.lr_dist2_workspace_bytes.argtypes = [ci]
        L.lr_dist2.restype = ci
        L.lr_dist2.argtypes = [ci, vp, vp, vp, vp]
        L.lr_views_workspace_bytes.restype = ctypes.c_size_t
        L.lr_views_workspace_bytes.argtypes = [ci, ci, ci, ll, ci]
        L.lr_views_accumulate.restype = ci
        L.lr_views_accumulate.argtypes = [ci, vp, vp, vp, vp, vp,                # n_views, view/proj/campos arrays, tanfovx/y arrays
                                          ci, ci, ci, vp, ci, ci,                # P D M bg W H
                                          vp, vp, vp, vp, vp, cf, vp, vp,        # means3D shs colors opac scales mod rot cov3D
                                          vp, vp, vp,                            # dL_dpix[], out_color[], out_radii[]
                                          vp, vp, vp, vp, vp, vp, vp, vp,        # 8 accumulators
                                          vp, ctypes.c_size_t, ll, ci, vp]       # workspace, bytes, capacity, n_streams, stream
        L.lr_views_check.restype = ci
        L.lr_views_check.argtypes = [vp, ci, ci, ci, ll, ci, vp]
        L.lr_views_train_workspace_bytes.restype = ctypes.c_size_t
        L.lr_views_train_workspace_bytes.argtypes = [ci, ci, ci, ll, ci]
        L.lr_views_train_accumulate.restype = ci
        L.lr_views_train_accumulate.argtypes = [ci, vp, vp, vp, vp, vp,          # n_views, view/proj/campos arrays, tanfovx/y arrays
                                                ci, ci, ci, vp, ci, ci,          # P D M bg W H
                                                vp, vp, vp, vp, cf, vp,          # means3D shs opac scales mod rot
                                                vp, cf, vp, vp, vp,              # targets[], lambda, out_losses, out_color[], out_radii[]
                                                vp, vp, vp, vp, vp, vp,          # 6 accumulators
                                                vp, ctypes.c_size_t, ll, ci, vp] # workspace, bytes, capacity, n_streams, stream
        L.lr_views_train_check.restype = ci
        L.lr_views_train_check.argtypes = [vp, ci, ci, ci, ll, ci, vp]
        L.lr_loss_workspace_bytes.restype = ctypes.c_size_t
        L.lr_loss_workspace_bytes.argtypes = [ci, ci, ci]
        L.lr_l1_dssim_forward.restype = ci
        L.lr_l1_dssim_forward.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, ctypes.c_size_t, vp]
        L.lr_l1_dssim_backward.restype = ci
        L.lr_l1_dssim_backward.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp]
        L.lr_l1_dssim_backward_weights.restype = ci
        L.lr_l1_dssim_backward_weights.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        L.lr_select_workspace_bytes.restype = ctypes.c_size_t
        L.lr_select_workspace_bytes.argtypes = [ci]
        L.lr_select_rows.restype = ci
        L.lr_select_rows.argtypes = [ci, vp, ci, vp, vp, vp, ll, vp, vp, ctypes.c_size_t, vp]
        L.lr_pack_ply_rows.restype = ci
        L.lr_pack_ply_rows.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        L.lr_densify_stats.restype = ci
        L.lr_densify_stats.argtypes = [ci, vp, vp, vp, vp, vp, vp]
        L.lr_adam_step.restype = ci
        cd = ctypes.c_double
        L.lr_adam_step.argtypes = [ci, vp, vp, vp, vp, vp, vp, cd, cd, cd, ci, vp]
        L.lr_step_begin.restype = ci
        L.lr_step_begin.argtypes = []
        L.lr_step_end.restype = ci
        L.lr_step_end.argtypes = [vp]
        L.lr_step_abort.restype = ci
        L.lr_step_abort.argtypes = []
        L.lr_forward_ticket.restype = ctypes.c_longlong
        L.lr_forward_ticket.argtypes = []
        L.lr_tune_set.restype = ci
        L.lr_tune_set.argtypes = [ctypes.c_char_p, ci]
        L.lr_last_launch_shapes.restype = ci
        L.lr_last_launch_shapes.argtypes = [ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.lr_profile_enable.restype = ci
        L.lr_profile_enable.argtypes = [ci]
        L.lr_profile_stage_name.restype = ctypes.c_char_p
        L.lr_profile_stage_name.argtypes = [ci]
        L.lr_profile_read.restype = ci
        L.lr_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ll), ci]
        _lib = L
    return _lib


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """`with on_device(dev):` -- make `dev` the current HIP device for the C call inside.  When it already is (the one-GPU case:
    every call of a training loop) this is a shared no-op object instead of torch.cuda.device's save / set / restore, ~8 us of
    host time per call on paths whose host time is what a LucidDreamer-sized iteration lasts."""
    import torch
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return _NO_GUARD if torch.cuda.current_device() == idx else torch.cuda.device(dev)


def tune_set(name, value):
    """Test hook: force one of the shipped code paths (lr_tune_set); value -1 restores the library's own rule.  Values that
    select a retired kernel raise unless the diagnostics build is loaded (diagnostics_build())."""
    rc = lib().lr_tune_set(name.encode(), int(value))
    if rc < 0:
        raise RuntimeError(last_error())


def diagnostics_build():
    """True when the loaded library was compiled with -DLR_DIAGNOSTICS (retired kernels, LR_* environment overrides)."""
    return b"+diagnostics" in lib().lr_version()


FWD_SHAPES = {-1: None, 0: "quadrant", 1: "quadrant-pairs", 2: "tile"}
BWD_SHAPES = {-1: None, 0: "half", 1: "quad", 2: "tile"}


def last_launch_shapes():
    """(forward, backward) kernel shapes of the process's last blend launches (lr_last_launch_shapes)."""
    f, b = ctypes.c_int(-1), ctypes.c_int(-1)
    lib().lr_last_launch_shapes(ctypes.byref(f), ctypes.byref(b))
    return FWD_SHAPES.get(f.value, f.value), BWD_SHAPES.get(b.value, b.value)


def profile_enable(on=True):
    """Start (clearing previous records) or stop per-stage HIP-event timing inside the library."""
    return lib().lr_profile_enable(1 if on else 0)


def profile_read():
    """{stage: (total_ms, calls)} for everything recorded since profile_enable(True); waits for the events."""
    L = lib()
    n = 16                                   # >= number of stages (lr_profile_read returns the real count)
    ms = (ctypes.c_double * n)()
    calls = (ctypes.c_longlong * n)()
    cnt = L.lr_profile_read(ms, calls, n)
    if cnt < 0:
        raise RuntimeError(last_error())
    return {L.lr_profile_stage_name(i).decode(): (ms[i], int(calls[i])) for i in range(cnt)}


def last_error():
    return lib().lr_last_error().decode("utf-8", "replace")


def raise_for(code, where):
    """Map a negative return code to the exception type the reference raises."""
    msg = last_error()
    if code == LR_ERR_INVALID_ARG:
        raise RuntimeError(f"{where}: {msg}")
    if code == LR_ERR_PREFILTERED:
        raise RuntimeError(msg)
    if code == LR_ERR_OVERFLOW:
        raise RuntimeError(f"{where}: {msg}")
    raise RuntimeError(f"{where}: error {code}: {msg}")
