"""Host-side (Python + launch) cost per view: tiny workload so the GPU is idle-bound; cProfile top functions."""
import cProfile, pstats, sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luciddreamer_amd import cameras, config, synthetic, parallel
from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
P, W, H = 2000, 1920, 1080
cloud = synthetic.make_cloud(P, "band", 0)
leaf = {k: v.to(dev).requires_grad_(True) for k, v in cloud.items()}
grads = parallel.FlatGrads(list(leaf.values()))
m2d = torch.zeros(P, 3, device=dev, requires_grad=True); m2d.grad = torch.zeros_like(m2d)
g = synthetic.upstream_grad(H, W).to(dev)
bg = torch.zeros(3, device=dev)
cams = [c.to(dev) for c in cameras.rotate360_path(W, H, n_views=30)]
rast = [GaussianRasterizer(GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), bg, 1.0,
        c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)) for c in cams]
config.set_async(True); config.set_fused_grad_accumulation(True)

def step():
    for r in rast:
        col, radii, dep = r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"],
                            scales=leaf["scales"], rotations=leaf["rotations"])
        col.backward(g)

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("host-bound us/view:", (time.perf_counter() - t0) / 300 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
