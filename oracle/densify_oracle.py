"""CPU oracle of GaussianModel's densify / prune / optimizer surgery and save_ply (TEST INFRASTRUCTURE ONLY).

Plain-torch restatement, on whatever device the tensors live on (tests use the CPU), of
/root/reference/scene/gaussian_model.py:
    _prune_optimizer :273-288, prune_points :290-304, cat_tensors_to_optimizer :306-326,
    densification_postfix :328-346, densify_and_split :348-373, densify_and_clone :375-389,
    densify_and_prune :391-403, save_ply attribute order :176-208,
and R/utils/general.py:78-100 build_rotation.  The "model" is a dict:
    params: {xyz,f_dc,f_rest,opacity,scaling,rotation}, exp_avg / exp_avg_sq: same keys (Adam moments),
    xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P], percent_dense.
Pinned by tests/test_densify_oracle_ref.py: the reference's GaussianModel is imported unchanged and run on CPU tensors
(oracle/ref_python.py maps its hard-coded device="cuda" to the CPU and stands in for plyfile / simple_knn); parameters,
both Adam moments and the statistics tensors agree bit for bit after prune / clone / split / densify_and_prune.
"""
import torch

KEYS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def prune_points(m, mask):
    valid = ~mask                                                       # :291
    for k in KEYS:                                                      # _prune_optimizer :273-288
        m["params"][k] = m["params"][k][valid]
        if m.get("exp_avg") is not None:
            m["exp_avg"][k] = m["exp_avg"][k][valid]
            m["exp_avg_sq"][k] = m["exp_avg_sq"][k][valid]
    m["xyz_gradient_accum"] = m["xyz_gradient_accum"][valid]            # :300-304
    m["denom"] = m["denom"][valid]
    m["max_radii2D"] = m["max_radii2D"][valid]


def densification_postfix(m, new):
    for k in KEYS:                                                      # cat_tensors_to_optimizer :306-326
        if m.get("exp_avg") is not None:
            m["exp_avg"][k] = torch.cat((m["exp_avg"][k], torch.zeros_like(new[k])), dim=0)
            m["exp_avg_sq"][k] = torch.cat((m["exp_avg_sq"][k], torch.zeros_like(new[k])), dim=0)
        m["params"][k] = torch.cat((m["params"][k], new[k]), dim=0)
    P = m["params"]["xyz"].shape[0]
    dev = m["params"]["xyz"].device
    m["xyz_gradient_accum"] = torch.zeros((P, 1), device=dev)           # :344-346
    m["denom"] = torch.zeros((P, 1), device=dev)
    m["max_radii2D"] = torch.zeros((P,), device=dev)


def build_rotation(r):
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_split(m, grads, grad_threshold, scene_extent, N=2, samples=None):
    p = m["params"]
    n_init = p["xyz"].shape[0]
    padded = torch.zeros((n_init,), device=p["xyz"].device)
    padded[:grads.shape[0]] = grads.squeeze()
    scaling = torch.exp(p["scaling"])
    sel = torch.where(padded >= grad_threshold, True, False)
    sel = torch.logical_and(sel, torch.max(scaling, dim=1).values > m["percent_dense"] * scene_extent)
    stds = scaling[sel].repeat(N, 1)
    if samples is None:
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
    rots = build_rotation(p["rotation"][sel]).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p["xyz"][sel].repeat(N, 1),
        "scaling": torch.log(scaling[sel].repeat(N, 1) / (0.8 * N)),
        "rotation": p["rotation"][sel].repeat(N, 1),
        "f_dc": p["f_dc"][sel].repeat(N, 1, 1),
        "f_rest": p["f_rest"][sel].repeat(N, 1, 1),
        "opacity": p["opacity"][sel].repeat(N, 1),
    }
    densification_postfix(m, new)
    prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=sel.device, dtype=torch.bool)))
    prune_points(m, prune_filter)
    return sel


def densify_and_clone(m, grads, grad_threshold, scene_extent):
    p = m["params"]
    sel = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
    sel = torch.logical_and(sel, torch.max(torch.exp(p["scaling"]), dim=1).values <= m["percent_dense"] * scene_extent)
    densification_postfix(m, {k: p[k][sel] for k in KEYS})
    return sel


def densify_and_prune(m, max_grad, min_opacity, extent, max_screen_size, split_samples_fn=None):
    grads = m["xyz_gradient_accum"] / m["denom"]
    grads[grads.isnan()] = 0.0
    densify_and_clone(m, grads, max_grad, extent)
    samples = split_samples_fn(m, grads, max_grad, extent) if split_samples_fn else None
    densify_and_split(m, grads, max_grad, extent, samples=samples)
    p = m["params"]
    prune_mask = (torch.sigmoid(p["opacity"]) < min_opacity).squeeze()
    if max_screen_size:
        big_vs = m["max_radii2D"] > max_screen_size
        big_ws = torch.exp(p["scaling"]).max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
    prune_points(m, prune_mask)


def ply_rows(m):
    """The attribute matrix of save_ply (:194-207): xyz, normals(0), f_dc, f_rest (both transposed to channel-major
    and flattened), opacity, scale, rotation."""
    p = m["params"]
    xyz = p["xyz"]
    f_dc = p["f_dc"].transpose(1, 2).flatten(start_dim=1)
    f_rest = p["f_rest"].transpose(1, 2).flatten(start_dim=1)
    return torch.cat((xyz, torch.zeros_like(xyz), f_dc, f_rest, p["opacity"], p["scaling"], p["rotation"]), dim=1)
