"""Runs the reference's own kernels (oracle/_ref gfx950 build) over the C3 workload: 30 rotate360 views forward+backward,
for a rocprofv3 kernel-trace of the reference on the MI355X (profiles/r02_reference_on_device_kernel_stats.md):

    rocprofv3 --kernel-trace --stats -d out -o ref -- python tests/ref_on_device_run.py
Test infrastructure (it drives oracle/), not a product path."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from luciddreamer_amd import cameras, synthetic      # noqa: E402
from oracle import ref_device                        # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cloud = {k: v.to(dev).contiguous() for k, v in synthetic.make_cloud(1_000_000, "band", 0).items()}
    cams = [c.to(dev) for c in cameras.rotate360_path(1920, 1080, n_views=30)]
    g = synthetic.upstream_grad(1080, 1920).to(dev)
    bg = torch.zeros(3, device=dev)
    r = ref_device.Renderer()

    def view(x):
        r.forward(bg, cloud["means3D"], None, cloud["opacities"], cloud["scales"], cloud["rotations"], 1.0, None,
                  x.world_view_transform.contiguous(), x.full_proj_transform.contiguous(), math.tan(x.FoVx * 0.5),
                  math.tan(x.FoVy * 0.5), 1080, 1920, cloud["shs"], 3, x.camera_center.contiguous(), sync=False)
        r.backward(g, sync=False)
    for x in cams[:3]:
        view(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in cams:
        view(x)
    ref_device.lib().refdev_sync()
    print(f"reference kernels on this GPU: {len(cams) / (time.perf_counter() - t0):.1f} views/s fwd+bwd (C3)")


if __name__ == "__main__":
    main()
