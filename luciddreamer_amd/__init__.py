"""MI355X-native differentiable Gaussian-splatting rasterizer (LucidDreamer hot path)."""
__version__ = "0.1.0"


def install(*args, **kwargs):
    """Switch an unchanged LucidDreamer caller onto the fused pieces of this library in one call (luciddreamer_amd/dropin.py)."""
    from .dropin import install as _install
    return _install(*args, **kwargs)


def uninstall(handle):
    from .dropin import uninstall as _uninstall
    return _uninstall(handle)
