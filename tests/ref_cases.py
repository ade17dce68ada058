"""Seeded input cases shared by tests/test_oracle_ref.py (restatement == oracle/_ref, bit for bit), by
tests/golden/make_ref_fixtures.py (committed outputs of oracle/_ref) and by the GPU parity tests against them.
Every case is a dict of numpy float32 arrays in the argument vocabulary of _C.rasterize_gaussians
(RAST/rasterize_points.h:18-38).  Built with numpy's PCG64 (bit-stable across numpy versions)."""
import math

import numpy as np

from luciddreamer_amd import cameras


def _cam(W, H, c2w=None):
    cam = cameras.make_camera(np.eye(4) if c2w is None else c2w, W, H)
    return dict(view=cam.world_view_transform.numpy(), proj=cam.full_proj_transform.numpy(),
                campos=cam.camera_center.numpy(), tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                W=W, H=H)


def _cloud(rng, P, lo, hi, s0, M=16, sigma=0.3):
    means = rng.uniform(lo, hi, size=(P, 3)).astype(np.float32)
    scales = np.exp(math.log(s0) + sigma * rng.standard_normal((P, 3))).astype(np.float32)
    q = rng.standard_normal((P, 4))
    rots = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-2.0 * rng.standard_normal((P, 1))))).astype(np.float32)
    shs = np.empty((P, M, 3), np.float32)
    shs[:, 0] = ((rng.uniform(size=(P, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
    if M > 1:
        shs[:, 1:] = (0.1 * rng.standard_normal((P, M - 1, 3))).astype(np.float32)
    return dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs)


def _case(name, cam, cloud, degree=3, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, colors_precomp=None,
          cov3D_precomp=None, grad_seed=1):
    rng = np.random.Generator(np.random.PCG64(grad_seed))
    c = dict(name=name, degree=degree, bg=np.asarray(bg, np.float32), scale_modifier=scale_modifier,
             colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp, **cam, **cloud)
    c["dL_dcolor"] = rng.standard_normal((3, cam["H"], cam["W"])).astype(np.float32)
    return c


def all_cases():
    R = lambda seed: np.random.Generator(np.random.PCG64(seed))
    cases = []
    # 1. plain box cloud in front of the camera, all SH degrees, non-zero background, ragged image size
    for D in range(4):
        cases.append(_case(f"box_deg{D}", _cam(200, 120), _cloud(R(10 + D), 3000, (-2, -1.2, 2.5), (2, 1.2, 5.5), 0.04),
                           degree=D, bg=(0.1, 0.3, 0.7)))
    # 2. Gaussians straddling the near plane z = 0.2 (auxiliary.h:154) and behind the camera
    cl = _cloud(R(20), 2000, (-0.3, -0.2, -0.5), (0.3, 0.2, 0.9), 0.01)
    cl["means3D"][:50, 2] = np.float32(0.2)                       # exactly on the plane: culled (<=)
    cl["means3D"][50:100, 2] = np.nextafter(np.float32(0.2), np.float32(1))
    cases.append(_case("near_plane", _cam(128, 96), cl, degree=2))
    # 3. far off-axis Gaussians: tx/tz beyond 1.3 tan(fov/2) -> clamped Jacobian, x_grad_mul/y_grad_mul = 0
    #    (forward.cu:82-87, backward.cu:175-176); large enough to still reach the image
    cl = _cloud(R(30), 1500, (-6, -4, 1.0), (6, 4, 4.0), 0.5, sigma=0.4)
    cases.append(_case("fov_clamp", _cam(160, 96), cl, degree=1, bg=(0.2, 0.2, 0.2)))
    # 4. depth ties: groups of Gaussians at identical means (identical depth bits) -> order by index
    cl = _cloud(R(40), 1200, (-1, -0.6, 3.0), (1, 0.6, 3.5), 0.05)
    cl["means3D"][:] = cl["means3D"][np.arange(1200) // 6 * 6]
    cl["means3D"][:, 2] = np.float32(3.25)
    cases.append(_case("depth_ties", _cam(96, 64), cl, degree=0))
    # 5. clamp mask: strongly negative DC so that many channels clamp at 0 (forward.cu:63-70, backward.cu:31-34)
    cl = _cloud(R(50), 2000, (-1.5, -1, 2.5), (1.5, 1, 5), 0.05)
    cl["shs"][:, 0] -= np.float32(1.0)
    cases.append(_case("sh_clamp", _cam(128, 80), cl, degree=3, bg=(0.5, 0.0, 0.25)))
    # 6. needle-like Gaussians close to the camera: a*c - b*b cancels (det == 0 / det < 0), radii cover every tile
    cl = _cloud(R(60), 400, (-0.4, -0.3, 0.3), (0.4, 0.3, 1.2), 0.02)
    cl["scales"][:, 0] *= np.float32(400.0)
    cl["scales"][:, 1:] *= np.float32(1e-4)
    cases.append(_case("needles", _cam(96, 64), cl, degree=1))
    # 7. precomputed colours and covariances (the other two input representations), scale_modifier != 1
    cl = _cloud(R(70), 2500, (-2, -1.2, 2.5), (2, 1.2, 5.5), 0.04)
    rng = R(71)
    A = (0.05 * rng.standard_normal((2500, 3, 3))).astype(np.float32)
    S = A @ A.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    cases.append(_case("colors_precomp", _cam(160, 100), cl, colors_precomp=rng.uniform(size=(2500, 3)).astype(np.float32),
                       scale_modifier=1.7, bg=(1.0, 1.0, 1.0)))
    cases.append(_case("cov3D_precomp", _cam(160, 100), cl, cov3D_precomp=cov, degree=2))
    # 8. rotated + translated camera (general view / projection matrices), opaque foreground -> early termination
    th = 0.4
    c2w = np.eye(4)
    c2w[:3, :3] = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    c2w[:3, 3] = (0.3, -0.1, -0.5)
    cl = _cloud(R(80), 4000, (-1, -1.2, 1.5), (3.5, 1.2, 5.5), 0.08)
    cl["opacities"][:] = np.float32(0.95)
    cases.append(_case("posed_opaque", _cam(176, 112, c2w), cl, degree=3))
    # 9. one Gaussian; nothing visible at all
    cl = _cloud(R(90), 1, (0, 0, 3), (0, 0, 3), 0.2)
    cases.append(_case("single", _cam(64, 48), cl, degree=0))
    cl = _cloud(R(91), 64, (-1, -1, -5), (1, 1, -1), 0.1)
    cases.append(_case("all_culled", _cam(64, 48), cl, degree=3))
    # 10. M = 1 (degree-0 model: sh tensor of shape (P,1,3))
    cases.append(_case("M1", _cam(96, 64), _cloud(R(100), 1000, (-1, -0.6, 2), (1, 0.6, 4), 0.05, M=1), degree=0))
    return cases


def forward_args(c):
    """Positional arguments of oracle.forward / ref.forward."""
    use_sh = c["colors_precomp"] is None
    use_cov = c["cov3D_precomp"] is not None
    return (c["bg"], c["means3D"], c["colors_precomp"], c["opacities"], None if use_cov else c["scales"],
            None if use_cov else c["rotations"], c["scale_modifier"], c["cov3D_precomp"], c["view"], c["proj"],
            c["tanfovx"], c["tanfovy"], c["H"], c["W"], c["shs"] if use_sh else None, c["degree"], c["campos"])
