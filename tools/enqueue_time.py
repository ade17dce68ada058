"""How long does the HOST need to enqueue one view (fwd+bwd) vs how long the GPU needs to execute it?"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luciddreamer_amd import cameras, config, synthetic, parallel
from luciddreamer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
P, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1920, 1080
cloud = synthetic.make_cloud(P, "band", 0)
leaf = {k: v.to(dev).requires_grad_(True) for k, v in cloud.items()}
grads = parallel.FlatGrads(list(leaf.values()))
m2d = torch.zeros(P, 3, device=dev, requires_grad=True); m2d.grad = torch.zeros_like(m2d)
g = synthetic.upstream_grad(H, W).to(dev); bg = torch.zeros(3, device=dev)
cams = [c.to(dev) for c in cameras.rotate360_path(W, H, n_views=30)]
rast = [GaussianRasterizer(GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), bg, 1.0,
        c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)) for c in cams]
config.set_async(True); config.set_fused_grad_accumulation(True)
for ns in (1, 2):
    pipe = parallel.ViewStreams(dev, ns)
    def step():
        pipe.begin_step()
        for r in rast:
            pipe.run_view(lambda r=r: r(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"],
                                        scales=leaf["scales"], rotations=leaf["rotations"])[0], lambda c: c.backward(g))
        pipe.end_step()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"streams={ns}: host enqueue {1e6*(t1-t0)/60:.1f} us/view, total {1e6*(t2-t0)/60:.1f} us/view")
