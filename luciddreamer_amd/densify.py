"""Densify / prune / optimizer-state surgery and .ply I/O for a GaussianModel on the MI355X.

Mirrors, method for method, what /root/reference/scene/gaussian_model.py does when the number of Gaussians changes:

    prune_points(model, mask)                                   :290-304   (+ _prune_optimizer :273-288)
    densification_postfix(model, new_xyz, ...)                  :328-346   (+ cat_tensors_to_optimizer :306-326)
    densify_and_clone(model, grads, grad_threshold, extent)     :375-389
    densify_and_split(model, grads, grad_threshold, extent, N)  :348-373
    densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size)   :391-403
    save_ply(model, path) / load_ply(model, path)               :193-208 / :215-256

`model` is duck-typed: the reference's GaussianModel works unchanged (attributes _xyz, _features_dc, _features_rest,
_opacity, _scaling, _rotation, optimizer with named param groups, xyz_gradient_accum, denom, max_radii2D,
percent_dense).  `patch(GaussianModel)` installs these functions as that class's methods.

How it differs from the reference inside: all row tensors (6 parameters, up to 12 Adam moments, 3 statistics) live in
capacity-sized buffers (RowStore); a prune is ONE order-preserving lr_select_rows call over all of them into a fresh set
(the old one is released afterwards), an append copies the selected rows behind the live ones, and parameters / Adam
moments become views of the store (six new nn.Parameter objects per change of size, no data copied for them).
No per-tensor boolean indexing, no torch.cat, no empty_cache().  The only host
synchronisation is reading the selected-row count (tensor shapes must be known to PyTorch).
Requires the HIP library (no CPU path).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

GROUP_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
              "scaling": "_scaling", "rotation": "_rotation"}
STAT_ATTRS = ("xyz_gradient_accum", "denom", "max_radii2D")


class RowStore:
    """Capacity-managed storage for a set of row tensors that always have the same number of rows.  One capacity-sized
    buffer per tensor; a compaction gathers into a freshly allocated set and releases the old one (so two copies
    exist only for the duration of a prune), an append writes behind the live rows of the same buffers."""

    def __init__(self, tensors, capacity=None, growth=1.5):
        self.P = int(next(iter(tensors.values())).shape[0])
        self.device = next(iter(tensors.values())).device
        if self.device.type != "cuda":
            raise RuntimeError("luciddreamer_amd.densify needs tensors on a HIP device (no CPU path)")
        self.growth = float(growth)
        self.cap = max(int(capacity or 0), int(self.P * self.growth) + 1024)
        self.row_shape, self.bufs = {}, {}
        for name, t in tensors.items():
            self._add(name, t)
        self._ws = None
        self._count = torch.zeros((1,), dtype=torch.int32, device=self.device)

    def _alloc(self, name, cap):
        return torch.zeros((cap,) + self.row_shape[name], dtype=torch.float32, device=self.device)

    def _add(self, name, t):
        if t.dtype != torch.float32 or int(t.shape[0]) != self.P:
            raise RuntimeError(f"RowStore: {name} must be float32 with {self.P} rows")
        self.row_shape[name] = tuple(t.shape[1:])
        self.bufs[name] = self._alloc(name, self.cap)
        self.bufs[name][:self.P].copy_(t.detach())

    def add_zero(self, name, row_shape):
        self.row_shape[name] = tuple(row_shape)
        self.bufs[name] = self._alloc(name, self.cap)

    def names(self):
        return list(self.bufs.keys())

    def view(self, name, start=0, stop=None):
        return self.bufs[name][start:self.P if stop is None else stop]

    def row_bytes(self, name):
        n = 1
        for d in self.row_shape[name]:
            n *= d
        return 4 * n

    def ensure_capacity(self, rows):
        if rows <= self.cap:
            return
        new_cap = max(int(rows * self.growth) + 1024, rows)
        for name, old in list(self.bufs.items()):
            fresh = self._alloc(name, new_cap)
            fresh[:self.P].copy_(old[:self.P])
            self.bufs[name] = fresh
        self.cap = new_cap

    def _select(self, mask_u8, n_rows, names, src, dst, dst_row_offset):
        """src / dst: {name: buffer}.  Zero-width tensors (e.g. features_rest of a degree-0 model, [P,0,3]) have
        nothing to move and are left out of the launch."""
        L = _lib.lib()
        names = [k for k in names if self.row_bytes(k) > 0]
        need = L.lr_select_workspace_bytes(n_rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        n = len(names)
        srcp = (ctypes.c_void_p * max(n, 1))(*[src[k].data_ptr() for k in names])
        dstp = (ctypes.c_void_p * max(n, 1))(*[dst[k].data_ptr() for k in names])
        rb = (ctypes.c_uint * max(n, 1))(*[self.row_bytes(k) for k in names])
        with torch.cuda.device(self.device):
            rc = L.lr_select_rows(n_rows, mask_u8.data_ptr(), n, srcp, dstp, rb, int(dst_row_offset), self._count.data_ptr(),
                                  self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        if rc < 0:
            _lib.raise_for(rc, "lr_select_rows")
        return int(self._count.item())                     # the one host sync: shapes must be known

    @staticmethod
    def _mask_u8(mask, n):
        m = mask.reshape(-1)
        if m.numel() != n:
            raise RuntimeError(f"mask has {m.numel()} entries, expected {n}")
        return m.to(torch.uint8).contiguous()

    def compact(self, keep_mask):
        """Keep the rows with keep_mask True, in order (all tensors, one call).  Returns the new row count."""
        m = self._mask_u8(keep_mask, self.P)
        fresh = {k: self._alloc(k, self.cap) for k in self.names()}
        count = self._select(m, self.P, self.names(), self.bufs, fresh, 0)
        self.bufs = fresh                                  # the old set is released once its last view is dropped
        self.P = count
        return count

    def append_selected(self, mask, names, repeat=1):
        """Copy the rows with mask True (mask over the first len(mask) rows) behind the live rows, `repeat` times
        (block-wise, like tensor[mask].repeat(N, ...)).  Tensors not in `names` get zero rows.  Returns (start, count)."""
        n_src = int(mask.numel())
        m = self._mask_u8(mask, n_src)
        count = self._select(m, n_src, [], self.bufs, self.bufs, 0)           # count only
        start = self.P
        self.ensure_capacity(self.P + repeat * count)
        if count:
            for r in range(repeat):
                self._select(m, n_src, list(names), self.bufs, self.bufs, start + r * count)
            for k in self.names():
                if k not in names:
                    self.bufs[k][start:start + repeat * count].zero_()
        self.P = start + repeat * count
        return start, count


# ------------------------------------------------------------------------------------------------------------
# binding a GaussianModel-like object to a store
# ------------------------------------------------------------------------------------------------------------
def _groups(model):
    return {g["name"]: g for g in model.optimizer.param_groups}


def _store(model):
    st = getattr(model, "_lr_store", None)
    params_now = {n: getattr(model, a) for n, a in GROUP_ATTR.items()}
    if st is not None and all(params_now[n].data_ptr() == st.view(n).data_ptr() and params_now[n].shape[0] == st.P
                              for n in GROUP_ATTR):
        # Adam creates its moments lazily: adopt them the first time they exist
        for name, g in _groups(model).items():
            state = model.optimizer.state.get(g["params"][0], None)
            if state is not None and "exp_avg" in state and name + ".exp_avg" not in st.bufs:
                for key in ("exp_avg", "exp_avg_sq"):
                    st.add_zero(f"{name}.{key}", st.row_shape[name])
                    st.view(f"{name}.{key}").copy_(state[key])
        _sync_stats_into_store(model, st)
        return st
    tensors = dict(params_now)
    for name, g in _groups(model).items():
        state = model.optimizer.state.get(g["params"][0], None)
        if state is not None and "exp_avg" in state:
            tensors[name + ".exp_avg"] = state["exp_avg"]
            tensors[name + ".exp_avg_sq"] = state["exp_avg_sq"]
    P = int(model._xyz.shape[0])
    dev = model._xyz.device
    for a in STAT_ATTRS:
        t = getattr(model, a, None)
        tensors[a] = t.to(torch.float32) if (t is not None and t.shape[0] == P) else torch.zeros(
            (P,) if a == "max_radii2D" else (P, 1), device=dev)
    st = RowStore({k: v.detach().contiguous() for k, v in tensors.items()})
    model._lr_store = st
    _bind(model, st)
    return st


def _sync_stats_into_store(model, st):
    for a in STAT_ATTRS:
        t = getattr(model, a, None)
        if t is not None and t.shape[0] == st.P and t.data_ptr() != st.view(a).data_ptr():
            st.view(a).copy_(t.to(torch.float32).view(st.view(a).shape))


def _bind(model, st):
    """Point parameters, Adam moments and statistics at the store's live rows.  A parameter whose row count changed
    becomes a NEW nn.Parameter over the store view (autograd caches a leaf's shape in its gradient accumulator, so
    re-pointing .data at a different shape is not safe); its optimizer state entry moves with it, as in
    _prune_optimizer / cat_tensors_to_optimizer (gaussian_model.py:273-326).  No tensor data is copied here."""
    groups = _groups(model)
    for name, attr in GROUP_ATTR.items():
        old = getattr(model, attr)
        view = st.view(name)
        group = groups.get(name)
        state = model.optimizer.state.pop(old, None) if group is not None else None
        if old.shape == view.shape:
            p = old
            p.data = view
            p.grad = None
        else:
            p = nn.Parameter(view.requires_grad_(True))
            setattr(model, attr, p)
            if group is not None:
                group["params"][0] = p
        if state is not None:
            if name + ".exp_avg" in st.bufs:
                state["exp_avg"] = st.view(name + ".exp_avg")
                state["exp_avg_sq"] = st.view(name + ".exp_avg_sq")
            model.optimizer.state[p] = state
    for a in STAT_ATTRS:
        setattr(model, a, st.view(a))


# ------------------------------------------------------------------------------------------------------------
# the reference's methods
# ------------------------------------------------------------------------------------------------------------
def add_densification_stats(model, viewspace_point_tensor, radii):
    """The per-iteration bookkeeping of the training loop in one kernel (R/luciddreamer.py:310-311 +
    GaussianModel.add_densification_stats, gaussian_model.py:405-407): for Gaussians with radii > 0,
    max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |viewspace grad[:, :2]|; denom += 1."""
    g = viewspace_point_tensor.grad
    P = int(radii.shape[0])
    ok = lambda t, n: t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    if g is None or not (ok(g, 3 * P) and ok(model.xyz_gradient_accum, P) and ok(model.denom, P) and ok(model.max_radii2D, P)) \
            or radii.dtype != torch.int32 or not radii.is_contiguous():
        raise RuntimeError("add_densification_stats needs contiguous float32 statistics / gradient and int32 radii on a HIP device")
    L = _lib.lib()
    dev = radii.device
    with _lib.on_device(dev):
        rc = L.lr_densify_stats(P, radii.data_ptr(), g.data_ptr(), model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr(),
                                model.max_radii2D.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if rc < 0:
        _lib.raise_for(rc, "lr_densify_stats")


def _no_fused_step_pending(model):
    """A backward that took the optimizer's step for its visible Gaussians (optim.FusedAdam.arm_fused_backward) must be
    followed by step() on the SAME parameter set: surgery in between is refused loudly (the reference would have densified on
    the pre-step parameters and skipped the step: the iteration must not have been armed)."""
    opt = getattr(model, "optimizer", None)
    if getattr(opt, "_fused_pending", None) is not None:
        raise RuntimeError("luciddreamer_amd.densify: the optimizer holds a fused step that step() has not finished; iterations "
                           "that densify / prune must not be armed (install(fuse_step=True) arms by the reference's schedule)")
    if hasattr(opt, "disarm"):
        opt.disarm()


def prune_points(model, mask):
    """Remove the Gaussians with mask True (gaussian_model.py:290-304)."""
    _no_fused_step_pending(model)
    st = _store(model)
    st.compact(~mask.reshape(-1).bool())
    _bind(model, st)


def densification_postfix(model, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation):
    """Append explicit new rows (gaussian_model.py:328-346); Adam moments of the new rows are zero, the three
    statistics are reset to zero for ALL rows, as in the reference."""
    _no_fused_step_pending(model)
    st = _store(model)
    n = int(new_xyz.shape[0])
    start = st.P
    st.ensure_capacity(start + n)
    new = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
           "scaling": new_scaling, "rotation": new_rotation}
    for k in st.names():
        tail = st.bufs[k][start:start + n]
        if k in new:
            tail.copy_(new[k].detach())
        else:
            tail.zero_()
    st.P = start + n
    for a in STAT_ATTRS:
        st.view(a).zero_()
    _bind(model, st)


def _get_scaling(model):
    return torch.exp(model._scaling)


def densify_and_clone(model, grads, grad_threshold, scene_extent):
    """gaussian_model.py:375-389: duplicate small Gaussians with a large view-space gradient."""
    st = _store(model)
    sel = torch.norm(grads, dim=-1) >= grad_threshold
    sel = torch.logical_and(sel, torch.max(_get_scaling(model), dim=1).values <= model.percent_dense * scene_extent)
    st.append_selected(sel, list(GROUP_ATTR.keys()))
    for a in STAT_ATTRS:
        st.view(a).zero_()
    _bind(model, st)


def _build_rotation(r):
    """R/utils/general.py:78-100 (normalises the quaternion first)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def densify_and_split(model, grads, grad_threshold, scene_extent, N=2, samples=None):
    """gaussian_model.py:348-373: replace large Gaussians with a large gradient by N smaller ones sampled from them.
    `samples` (optional, [N*n_selected, 3] standard-normal draws scaled by the selected scales) overrides the
    torch.normal call so that tests can feed both implementations the same random numbers."""
    st = _store(model)
    n_init = st.P
    padded = torch.zeros((n_init,), device=model._xyz.device)
    padded[:grads.shape[0]] = grads.squeeze()
    scaling = _get_scaling(model)
    sel = padded >= grad_threshold
    sel = torch.logical_and(sel, torch.max(scaling, dim=1).values > model.percent_dense * scene_extent)
    stds = scaling[sel].repeat(N, 1)
    if samples is None:
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
    rots = _build_rotation(model._rotation[sel]).repeat(N, 1, 1)
    new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + model._xyz[sel].repeat(N, 1)
    new_scaling = torch.log(scaling[sel].repeat(N, 1) / (0.8 * N))
    # rotation, SH and opacity rows are copied by the row kernel (N blocks); xyz and scaling are overwritten
    start, count = st.append_selected(sel, list(GROUP_ATTR.keys()), repeat=N)
    if count:
        st.bufs["xyz"][start:start + N * count].copy_(new_xyz)
        st.bufs["scaling"][start:start + N * count].copy_(new_scaling)
    for a in STAT_ATTRS:
        st.view(a).zero_()
    _bind(model, st)
    prune_filter = torch.cat((sel, torch.zeros(N * count, device=sel.device, dtype=torch.bool)))
    prune_points(model, prune_filter)


def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size):
    """gaussian_model.py:391-403."""
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    densify_and_clone(model, grads, max_grad, extent)
    densify_and_split(model, grads, max_grad, extent)
    prune_mask = (torch.sigmoid(model._opacity) < min_opacity).squeeze()
    if max_screen_size:
        big_points_vs = model.max_radii2D > max_screen_size
        big_points_ws = _get_scaling(model).max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
    prune_points(model, prune_mask)


# ------------------------------------------------------------------------------------------------------------
# .ply (binary little endian, the layout plyfile writes for GaussianModel.save_ply)
# ------------------------------------------------------------------------------------------------------------
def ply_attribute_names(n_rest):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * n_rest)]
    return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_ply(model, path):
    """gaussian_model.py:193-208, without the per-vertex Python tuples: the 62-float records are assembled by one
    HIP kernel and copied to the host once."""
    L = _lib.lib()
    dev = model._xyz.device
    P = int(model._xyz.shape[0])
    n_rest = int(model._features_rest.shape[1])
    props = 17 + 3 * n_rest
    rows = torch.empty((P, props), dtype=torch.float32, device=dev)
    c = lambda t: t.detach().contiguous()
    t = [c(model._xyz), c(model._features_dc), c(model._features_rest), c(model._opacity), c(model._scaling), c(model._rotation)]
    with torch.cuda.device(dev):
        rc = L.lr_pack_ply_rows(P, 1 + n_rest, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr() if n_rest else None,
                                t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), rows.data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
    if rc < 0:
        _lib.raise_for(rc, "lr_pack_ply_rows")
    host = rows.cpu().numpy()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in ply_attribute_names(n_rest)) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(host.astype("<f4", copy=False).tobytes())


def read_ply(path):
    """{property name: float32 array} of the vertex element of a binary-little-endian or ascii float .ply."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise RuntimeError(f"{path}: not a ply file")
        fmt, n, names, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise RuntimeError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise RuntimeError(f"{path}: only float32 vertex properties are supported (got {tok[1]})")
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(4 * n * len(names)), dtype="<f4").reshape(n, len(names))
        elif fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float32, max_rows=n).reshape(n, len(names))
        else:
            raise RuntimeError(f"{path}: unsupported ply format {fmt}")
    return {k: np.ascontiguousarray(data[:, i]) for i, k in enumerate(names)}


def load_ply(model, path, device="cuda"):
    """gaussian_model.py:215-256: fills the six parameters from a .ply written by save_ply."""
    v = read_ply(path)
    col = lambda prefix: sorted((k for k in v if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    f_dc = np.stack([v[f"f_dc_{i}"] for i in range(3)], axis=1)[:, None, :]                  # [P,1,3]
    rest = col("f_rest_")
    n_rest = len(rest) // 3
    max_deg = getattr(model, "max_sh_degree", None)
    if max_deg is not None and len(rest) != 3 * (max_deg + 1) ** 2 - 3:
        raise AssertionError("f_rest count does not match max_sh_degree")                   # :232
    f_rest = np.stack([v[k] for k in rest], axis=1).reshape(-1, 3, n_rest).transpose(0, 2, 1) if n_rest else \
        np.zeros((xyz.shape[0], 0, 3), np.float32)                                          # channel-major -> [P,n_rest,3]
    mk = lambda a: nn.Parameter(torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=device).requires_grad_(True))
    model._xyz = mk(xyz)
    model._features_dc = mk(f_dc)
    model._features_rest = mk(f_rest)
    model._opacity = mk(v["opacity"][:, None])
    model._scaling = mk(np.stack([v[k] for k in col("scale_")], axis=1))
    model._rotation = mk(np.stack([v[k] for k in col("rot_")], axis=1))
    if max_deg is not None:
        model.active_sh_degree = max_deg
    if hasattr(model, "_lr_store"):
        del model._lr_store


def patch(cls):
    """Install the functions above as methods of a GaussianModel class (the reference's, unchanged otherwise)."""
    for fn in (prune_points, densification_postfix, densify_and_clone, densify_and_split, densify_and_prune, save_ply):
        setattr(cls, fn.__name__, fn)
    cls.load_ply = lambda self, path: load_ply(self, path)
    return cls
