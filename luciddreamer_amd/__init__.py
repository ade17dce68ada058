"""MI355X-native differentiable Gaussian-splatting rasterizer (LucidDreamer hot path)."""
__version__ = "0.1.0"
