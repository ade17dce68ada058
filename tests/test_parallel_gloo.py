"""Multi-process (gloo, world_size 2, CPU) tests of the view-sharded data-parallel step.

The product rasterizer has no CPU path, so the per-view loss here renders through the float64 PyTorch
restatement (oracle/torch_oracle.py, test infrastructure) -- what is under test is the sharding, the
flat gradient bucket and the single all-reduce: n ranks must produce exactly the gradient one rank
produces over all views (up to fp32 reduction order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from luciddreamer_amd import cameras, parallel, synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_problem():
    cloud = synthetic.make_cloud(60, "band", 2, scale_mult=0.2)
    cams = cameras.rotate360_path(32, 32, n_views=5)
    return cloud, cams


def _params(cloud):
    return [cloud[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")]


def _loss_fn(params, H=32, W=32):
    from oracle import torch_oracle
    from tests import helpers as hp
    means3D, scales, rotations, opacities, shs = params
    g = synthetic.upstream_grad(H, W).double()

    def loss(cam, i):
        tfx, tfy = hp.tan_fov(cam)
        col, _, _ = torch_oracle.render(means3D, opacities, cam.world_view_transform, cam.full_proj_transform,
                                        cam.camera_center, tfx, tfy, H, W, torch.zeros(3), scales=scales,
                                        rotations=rotations, shs=shs, degree=2)
        return (col * g).sum().float() * (1.0 + 0.1 * i)
    return loss


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = parallel.init_distributed(backend="gloo", device=torch.device("cpu"))
    assert (r, w) == (rank, world) and parallel.world_size() == world and parallel.get_rank() == rank
    cloud, cams = _make_problem()
    params = _params(cloud)
    grads = parallel.dp_step(cams, params, _loss_fn(params))
    # a second step re-uses the bucket (zeroed, p.grad still views into it)
    grads = parallel.dp_step(cams, params, _loss_fn(params), grads=grads)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, grads.views))
    # densification statistics: SUM, SUM, MAX
    acc = torch.full((4, 1), float(rank + 1))
    den = torch.full((4, 1), 1.0)
    mx = torch.tensor([float(rank), 5.0 - rank, 2.0, 7.0 * rank])
    parallel.all_reduce_densification_stats(acc, den, mx)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=grads.flat.numpy(), acc=acc.numpy(), den=den.numpy(), mx=mx.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_views_partition():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += parallel.shard_views(30, r, world)
        assert sorted(seen) == list(range(30))
    assert parallel.shard_views(30, 1, 8) == [1, 9, 17, 25]


def test_flat_grads_bucket_is_contiguous_and_59_floats_per_gaussian():
    cloud, _ = _make_problem()
    params = _params(cloud)
    fg = parallel.FlatGrads(params)
    assert fg.flat.numel() == 59 * 60 and fg.flat.is_contiguous()
    off = 0
    for p in params:
        assert p.grad.data_ptr() == fg.flat.data_ptr() + 4 * off and p.grad.shape == p.shape
        off += p.numel()


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_rank(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]), "replicas diverged after the all-reduce"
    # single process, all views
    cloud, cams = _make_problem()
    params = _params(cloud)
    ref = parallel.dp_step(cams, params, _loss_fn(params), rank=0, world=1, reduce=False)
    a, b = r0["flat"], ref.flat.numpy()
    assert np.abs(b).max() > 0
    assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    assert np.array_equal(r0["acc"], np.full((4, 1), 3.0)) and np.array_equal(r0["den"], np.full((4, 1), 2.0))
    assert np.array_equal(r0["mx"], np.array([1.0, 5.0, 2.0, 7.0])) and np.array_equal(r1["mx"], r0["mx"])


# ---------------------------------------------------------------------------------------------------------------------
# ShardedAdam: reduce-scatter + Adam on the rank's shard + all-gather of the parameters  ==  all-reduce + Adam everywhere
# ---------------------------------------------------------------------------------------------------------------------
def _torch_adam_pieces(pieces, beta1, beta2, eps, step):
    """torch.optim.Adam's single-tensor arithmetic on slices (the CPU stand-in for lr_adam_step in this test: what is
    under test is the sharding and the two collectives, not the kernel -- tests/test_gpu_optim.py covers that)."""
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    for p, g, m, v, lr in pieces:
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


LRS = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3]


def _fake_view_grads(params, rank, world, n_views, it):
    """Deterministic stand-ins for per-view gradients: view i contributes a fixed pseudo-random tensor."""
    out = [torch.zeros_like(p) for p in params]
    for i in parallel.shard_views(n_views, rank, world):
        g = torch.Generator().manual_seed(1000 * it + i)
        for o in out:
            o.add_(torch.randn(o.shape, generator=g))
    return out


def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo", device=torch.device("cpu"))
    cloud, _ = _make_problem()
    params = [torch.nn.Parameter(cloud[k].clone()) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    grads = parallel.ShardedAdam.make_buckets(params)
    opt = parallel.ShardedAdam(params, grads, LRS, adam_fn=_torch_adam_pieces)
    assert grads.flat.numel() % (4 * world) == 0 and opt.shard * world == grads.flat.numel()
    assert all(p.data_ptr() == v.data_ptr() for p, v in zip(params, opt.params.views))      # parameters live in the flat buffer
    for it in range(3):
        grads.zero_()
        for p, g in zip(params, _fake_view_grads(params, rank, world, 7, it)):
            p.grad.add_(g)
        opt.lrs[0] = LRS[0] * (0.9 ** it)                                                  # a schedule on one tensor
        opt.step()
    np.savez(os.path.join(out_dir, f"sh{rank}.npz"), flat=opt.params.flat.numpy(), m=opt.exp_avg.numpy(),
             lo=opt.lo, hi=opt.hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_adam_two_ranks_equal_unsharded(tmp_path):
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "sh0.npz"), np.load(tmp_path / "sh1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]), "replicas diverged after the all-gather"
    assert int(r0["lo"]) == 0 and int(r0["hi"]) == int(r1["lo"])                               # the shards tile the bucket
    # one process: all views, torch.optim.Adam with per-tensor learning rates
    cloud, _ = _make_problem()
    params = [torch.nn.Parameter(cloud[k].clone()) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(params, LRS)], lr=0.0, eps=1e-15)
    for it in range(3):
        ref.param_groups[0]["lr"] = LRS[0] * (0.9 ** it)
        for p, g in zip(params, _fake_view_grads(params, 0, 1, 7, it)):
            p.grad = g
        ref.step()
    layout = parallel.FlatGrads([torch.zeros_like(p) for p in params], multiple_of=8)
    got = r0["flat"]
    for p, (off, n) in zip(params, layout.segments):
        a, b = got[off:off + n], p.detach().numpy().ravel()
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(b).max()), float(np.abs(a - b).max())


def test_sharded_adam_single_rank_is_plain_adam():
    cloud, _ = _make_problem()
    params = [torch.nn.Parameter(cloud[k].clone()) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    grads = parallel.ShardedAdam.make_buckets(params)
    opt = parallel.ShardedAdam(params, grads, LRS, adam_fn=_torch_adam_pieces)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(twin, LRS)], lr=0.0, eps=1e-15)
    for it in range(2):
        grads.zero_()
        gs = _fake_view_grads(params, 0, 1, 3, it)
        for p, q, g in zip(params, twin, gs):
            p.grad.add_(g)
            q.grad = g.clone()
        opt.step()
        ref.step()
    for p, q in zip(params, twin):
        assert torch.allclose(p, q, rtol=0, atol=2e-6 * max(1.0, float(q.abs().max())))
    with pytest.raises(ValueError):
        parallel.ShardedAdam(params, parallel.FlatGrads(params, multiple_of=1) if grads.flat.numel() % 4 else grads, LRS[:2])
