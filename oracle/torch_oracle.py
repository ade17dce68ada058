"""Pure-PyTorch differentiable restatement of the reference rasterizer (float64 by default).

TEST INFRASTRUCTURE ONLY (see raster_oracle.c header).  PARITY UNPINNED.

Purpose: an *independent* statement of the forward semantics whose AUTOGRAD gradients
cross-check the hand-written backward formulas restated in raster_oracle.c (which follow
RAST/cuda_rasterizer/backward.cu).  Three places where the reference's backward is
deliberately NOT the true derivative are reproduced with detach() tricks:
  (i)  alpha = min(0.99, o*G) is differentiated as if unclamped (backward.cu:513, 567);
  (ii) the fov clamp of t.x/t.z, t.y/t.z: inside the range t is used as is, outside it is a
       constant (x_grad_mul/y_grad_mul, backward.cu:175-176, 262-264);
  (iii) the depth output carries no gradient (backward.cu:457-464, 539-554 commented out).
Small inputs only (python loop over tiles, dense [pixels x gaussians] per tile).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    """RAST/cuda_rasterizer/forward.cu:20-71.  sh: (P,M,3), dirs: (P,3) normalised."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def cov3d_from_scale_rot(scales, mod, rot):
    """RAST/cuda_rasterizer/forward.cu:118-152: Sigma = R S S R^T, quaternion (r,x,y,z) as given."""
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    s = mod * scales
    M = R * s[:, None, :]          # R @ diag(s)
    Sigma = M @ M.transpose(1, 2)
    return torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2],
                        Sigma[:, 2, 2]], dim=1)


def render(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, bg,
           scales=None, rotations=None, scale_modifier=1.0, cov3D_precomp=None,
           shs=None, degree=0, colors_precomp=None, means2D=None, dtype=torch.float64):
    """Returns (color (3,H,W), depth (1,H,W) [no grad], radii (P,) int32)."""
    cv = lambda t: None if t is None else t.to(dtype)
    means3D, opacities, scales, rotations = cv(means3D), cv(opacities), cv(scales), cv(rotations)
    cov3D_precomp, shs, colors_precomp, means2D = cv(cov3D_precomp), cv(shs), cv(colors_precomp), cv(means2D)
    V, Pm, campos, bg = cv(viewmatrix), cv(projmatrix), cv(campos), cv(bg)
    P = means3D.shape[0]
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    gx, gy = (W + 15) // 16, (H + 15) // 16

    p_view = means3D @ V[:3, :3] + V[3, :3]                       # auxiliary.h:58-66
    hom = means3D @ Pm[:3, :] + Pm[3, :]                           # auxiliary.h:68-77
    p_w = 1.0 / (hom[:, 3] + 0.0000001)
    ndc = hom[:, :3] * p_w[:, None]
    if means2D is not None:                                        # virtual screen-space offsets (NDC units)
        ndc = torch.cat([ndc[:, :2] + means2D[:, :2], ndc[:, 2:3]], dim=1)
    in_front = p_view[:, 2] > 0.2                                  # auxiliary.h:154

    cov3D = cov3D_precomp if cov3D_precomp is not None else cov3d_from_scale_rot(scales, scale_modifier, rotations)

    # forward.cu:74-113 with the backward's clamp convention (module docstring (ii))
    tz = p_view[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], dim=1).reshape(P, 2, 3)
    Wc = V[:3, :3].T                                               # world->camera rotation
    A = J @ Wc
    Sig = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                       cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], dim=1).reshape(P, 3, 3)
    cov2 = A @ Sig @ A.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5                       # auxiliary.h:41-44
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    ri = radius.to(torch.int64).to(dtype)
    trunc = lambda t: torch.trunc(t.detach()).to(torch.int64)      # C (int) cast
    rminx = trunc((px - ri) / 16).clamp(0, gx); rmaxx = trunc((px + ri + 15) / 16).clamp(0, gx)
    rminy = trunc((py - ri) / 16).clamp(0, gy); rmaxy = trunc((py + ri + 15) / 16).clamp(0, gy)
    ok = ok & (((rmaxx - rminx) * (rmaxy - rminy)) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        rgb = sh_to_rgb(degree, shs, d / d.norm(dim=1, keepdim=True))
    depth_g = p_view[:, 2].detach()
    op = opacities.reshape(-1)

    color = torch.zeros(3, H, W, dtype=dtype)
    depth = torch.zeros(1, H, W, dtype=dtype)
    color_rows = []
    # stable order by (float32 depth bits, index): rasterizer_impl.cu:98-108, 304-309
    order_all = torch.argsort(depth_g.to(torch.float32), stable=True)
    out_c = [[None] * gx for _ in range(gy)]
    out_d = [[None] * gx for _ in range(gy)]
    for ty_ in range(gy):
        for tx_ in range(gx):
            sel = ok & (rminx <= tx_) & (tx_ < rmaxx) & (rminy <= ty_) & (ty_ < rmaxy)
            ids = order_all[sel[order_all]]
            ys, xs = torch.meshgrid(torch.arange(ty_ * 16, ty_ * 16 + 16), torch.arange(tx_ * 16, tx_ * 16 + 16),
                                    indexing="ij")
            pxf, pyf = xs.reshape(-1).to(dtype), ys.reshape(-1).to(dtype)
            if ids.numel() == 0:
                c_t = bg[:, None].expand(3, 256)
                d_t = torch.zeros(256, dtype=dtype)
            else:
                dx = px[ids][None, :] - pxf[:, None]
                dy = py[ids][None, :] - pyf[:, None]
                cn = conic[ids]
                power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
                G = torch.exp(torch.clamp_max(power, 0.0))
                araw = op[ids][None, :] * G
                alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()       # docstring (i)
                valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
                a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                one_m = 1.0 - a_eff
                T_incl = torch.cumprod(one_m, dim=1)
                T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
                stop = valid & (T_incl.detach() < 0.0001)
                done = torch.cumsum(stop.to(torch.int64), dim=1) > 0                # inclusive: stopper not applied
                contrib = valid & ~done
                w = torch.where(contrib, alpha * T_excl, torch.zeros_like(alpha))
                # T after the last applied Gaussian
                T_fin = torch.prod(torch.where(contrib, one_m, torch.ones_like(one_m)), dim=1)
                c_t = (w @ rgb[ids]).T + T_fin[None, :] * bg[:, None]
                acc = 0.000001 + w.sum(dim=1)
                Dacc = w @ depth_g[ids]
                d_t = torch.where(acc > 0.5, Dacc / acc, torch.zeros_like(acc)).detach()   # docstring (iii)
            out_c[ty_][tx_] = c_t.reshape(3, 16, 16)
            out_d[ty_][tx_] = d_t.reshape(1, 16, 16)
    color = torch.cat([torch.cat(row, dim=2) for row in out_c], dim=1)[:, :H, :W]
    depth = torch.cat([torch.cat(row, dim=2) for row in out_d], dim=1)[:, :H, :W]
    return color, depth, radii
