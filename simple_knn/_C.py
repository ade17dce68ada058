"""`simple_knn._C.distCUDA2` (reference: /root/reference/submodules/simple-knn/ext.cpp:15-17,
spatial.cu:15-26) on the MI355X library (lr_dist2 in include/lucid_raster.h)."""
import torch

from luciddreamer_amd import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points (P,3) float32 on a HIP device -> (P,) mean squared distance to the 3 nearest other points."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a HIP device (no CPU path)")
    if points.dtype != torch.float32:
        raise RuntimeError("distCUDA2: points must be float32")
    pts = points.contiguous()
    P = int(pts.size(0))
    out = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = _lib.lib()
    ws = torch.empty((L.lr_dist2_workspace_bytes(P),), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = L.lr_dist2(P, pts.data_ptr(), out.data_ptr(), ws.data_ptr(),
                        torch.cuda.current_stream(pts.device).cuda_stream)
    if rc < 0:
        _lib.raise_for(rc, "distCUDA2")
    return out
