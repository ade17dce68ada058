"""Worker of tests/test_gpu_distributed.py: one rank of a data-parallel multi-view step on the real HIP path.
Launched by torch.distributed.run with LR_DIST_BACKEND=gloo so that several ranks can share the one GPU of a gpurun box
(RCCL refuses two ranks on one device; gloo stages the same all_reduce calls through the host).  Rank 0 writes the reduced
gradient bucket to argv[1]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from luciddreamer_amd import cameras, parallel, synthetic       # noqa: E402


LRS = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3]          # means3D, scales, rotations, opacity, sh (ChunkedViewStep.ORDER)


def build(dev, n_views, world, rank, chunks, sharded=False):
    cloud = {k: torch.nn.Parameter(v.to(dev)) for k, v in synthetic.make_cloud(30_000, "band", 6).items()}
    path = cameras.rotate360_path(256, 160, n_views=n_views)
    mine = [path[i].to(dev) for i in parallel.shard_views(n_views, rank, world)]
    g = synthetic.upstream_grad(160, 256).to(dev)
    named = {"means3D": cloud["means3D"], "scales": cloud["scales"], "rotations": cloud["rotations"],
             "opacity": cloud["opacities"], "sh": cloud["shs"]}
    ordered = [named[k] for k in parallel.ChunkedViewStep.ORDER]
    grads = parallel.ShardedAdam.make_buckets(ordered) if sharded else None
    step = parallel.ChunkedViewStep(mine, [g] * len(mine), named, 3, torch.zeros(3, device=dev), 400_000, n_streams=2,
                                    chunks=chunks, grads=grads)
    if sharded:
        return step, torch.zeros(30_000, 3, device=dev), parallel.ShardedAdam(ordered, step.grads, LRS)
    return step, torch.zeros(30_000, 3, device=dev)


def sharded_main(out):
    """BASELINE.json config 3's step as a training step: the 8 views of a step shared by the ranks, gradients exchanged
    by reduce-scatter, Adam on each rank's shard (the HIP kernel), parameters all-gathered."""
    rank, world, dev = parallel.init_distributed()
    step, m2d, opt = build(dev, 8, world, rank, 1, sharded=True)
    for _ in range(3):
        step.run(m2d, reduce=False)
        opt.step()
    step.check()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": opt.params.flat.cpu(), "segments": step.grads.segments, "world": world}, out)
    if world > 1:
        torch.distributed.destroy_process_group()


def sparse_main(out):
    """The metric's step with the reduce half of the exchange as an all-to-all of the touched rows."""
    rank, world, dev = parallel.init_distributed()
    step, m2d = build(dev, 8, world, rank, 1)
    infos = []
    for _ in range(2):
        step.run(m2d, reduce=False)
        infos.append(parallel.sparse_rows_all_reduce(step.grads.views))
    step.check()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"flat": step.grads.flat.cpu(), "world": world, "info": infos[-1]}, out)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    out = sys.argv[1]
    rank, world, dev = parallel.init_distributed()
    step, m2d = build(dev, 8, world, rank, None)
    assert len(step.buckets) == (parallel.REDUCE_CHUNKS if world > 1 else 1)
    for _ in range(2):                                   # twice: buckets are re-zeroed, works re-issued
        step.run(m2d)
    step.check()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"flat": step.grads.flat.cpu(), "world": world, "backend": torch.distributed.get_backend() if world > 1 else None},
                   out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "sharded":
        sharded_main(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "sparse":
        sparse_main(sys.argv[1])
    else:
        main()
