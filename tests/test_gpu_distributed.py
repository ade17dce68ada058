"""GPU: the data-parallel path on the real HIP kernels with more than one rank.  A gpurun box has ONE MI355X, and RCCL
refuses two ranks on one device, so the ranks talk over gloo (LR_DIST_BACKEND=gloo): same torch.distributed calls, same
sharding, same overlapped per-chunk all-reduce (parallel.ChunkedViewStep), only the transport differs.  RCCL itself can
only run on a multi-GPU node (the driver's SCALE run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(n, script_args, port, extra_env=None, timeout=600):
    env = dict(os.environ, LR_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_equal_one_rank_on_the_flat_gradient_bucket(hip_device, tmp_path):
    from tests import dist_gpu_worker as W
    step, m2d = W.build(hip_device, 8, 1, 0, 1)          # all 8 views on one rank, one bucket, no collective
    step.run(m2d)
    step.check()
    want = step.grads.flat.cpu().numpy()
    out = str(tmp_path / "flat.pt")
    r = _torchrun(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out], 29611)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["world"] == 2 and got["backend"] == "gloo"
    a = got["flat"].numpy()
    assert a.shape == want.shape and np.abs(want).max() > 0
    assert np.abs(a - want).max() <= 2e-5 * np.abs(want).max()       # same sum, different association across ranks / chunks


def test_sparse_rows_exchange_two_ranks_equal_one_rank(hip_device, tmp_path):
    """--exchange sparse-rows on the real kernels: the rows each rank's 4 views touched go to their owners (all-to-all), the
    owners' blocks are all-gathered; the bucket must equal one rank's sum over all 8 views, and the rows sent must be a
    fraction of the scene."""
    from tests import dist_gpu_worker as W
    step, m2d = W.build(hip_device, 8, 1, 0, 1)
    step.run(m2d)
    step.check()
    want = step.grads.flat.cpu().numpy()
    out = str(tmp_path / "sparse.pt")
    r = _torchrun(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, "sparse"], 29619)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    a = got["flat"].numpy()
    assert got["world"] == 2 and a.shape == want.shape and np.abs(want).max() > 0
    assert np.abs(a - want).max() <= 2e-5 * np.abs(want).max()
    info = got["info"]
    assert 0 < info["sent_rows"] < 15_000 and info["bytes_all_to_all"] < info["dense_equivalent_bytes"] // 2, info


def test_sharded_adam_two_ranks_equal_one_rank_with_fused_adam(hip_device, tmp_path):
    """The strong-scaling training step on the real kernels: 8 views per step shared by 2 ranks, reduce-scatter of the
    flat bucket, lr_adam_step on each rank's shard, all-gather of the parameters -- against ONE rank rendering all 8 views
    and stepping optim.FusedAdam (torch.optim.Adam's arithmetic) on the whole parameters, 3 steps."""
    from luciddreamer_amd.optim import FusedAdam
    from tests import dist_gpu_worker as W
    step, m2d = W.build(hip_device, 8, 1, 0, 1)
    ordered = [step.named[k] for k in step.ORDER]
    ref = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ordered, W.LRS)], lr=0.0, eps=1e-15)
    for _ in range(3):
        step.run(m2d)
        ref.step()
    step.check()
    out = str(tmp_path / "sharded.pt")
    r = _torchrun(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, "sharded"], 29617)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["world"] == 2
    flat = got["params"].numpy()
    for p, (off, n) in zip(ordered, got["segments"]):
        a, b = flat[off:off + n], p.detach().cpu().numpy().ravel()
        # same Adam arithmetic on gradients that differ by float summation order across the ranks; Adam's eps = 1e-15
        # turns a sign flip of a float-noise gradient into a +-lr step, so the bar is a few learning-rate steps
        assert np.abs(a - b).max() <= 3 * 3 * 5e-2 * 1.001 and np.median(np.abs(a - b)) <= 1e-6, (float(np.abs(a - b).max()),)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_runs_under_two_ranks_and_reports_the_world_size(hip_device, scaling):
    """N > 1 defaults to the stated configuration (BASELINE.json config 3): the step's views are SHARED by the ranks, one
    all-reduce per step ("strong"); --scaling weak keeps that many views per rank."""
    extra = [] if scaling == "strong" else ["--scaling", "weak"]
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gaussians", "100000",
                      "--views", "4", "--resolution", "640x360", "--no-cpu-baseline", "--sustain-seconds", "0"] + extra,
                  29613 if scaling == "strong" else 29615)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["dist_world_size"] == 2 and line["config"]["dist_backend"] == "gloo"
    assert line["scaling"] == scaling and line["value"] > 0 and line["entry_points"] is None
    cfg = line["config"]
    assert cfg["views_per_step"] == (4 if scaling == "strong" else 8) and cfg["views_per_rank_per_step"] == (2 if scaling == "strong" else 4)
    assert cfg["allreduce_bytes_per_step"] == cfg["grad_bucket_bytes"]          # few views per rank: one all-reduce per step


def test_bench_reports_bytes_on_wire_for_both_exchanges(hip_device):
    """config.allreduce_bytes_per_step for N > 1: the ring model for the dense all-reduce, measured rows for --exchange
    sparse-rows."""
    lines = {}
    for ex, port in (("allreduce", 29621), ("sparse-rows", 29623)):
        r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "100000",
                          "--views", "4", "--resolution", "640x360", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extras",
                          "--exchange", ex], port)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines[ex] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["config"]
    dense, sparse = lines["allreduce"], lines["sparse-rows"]
    assert dense["allreduce_bytes_per_step"] == 2 * 1 * dense["allreduce_payload_bytes_per_step"] // 2
    d = sparse["exchange_detail"]
    assert sparse["exchange"] == "sparse-rows" and d["rows_sent_per_step"] > 0
    assert sparse["allreduce_bytes_per_step"] == d["bytes_all_to_all"] + d["bytes_all_gather"]
    assert d["bytes_all_to_all"] < d["dense_ring_allreduce_bytes"] // 2          # the reduce half shrank


def test_bench_gpus_n_starts_its_own_ranks_or_refuses(hip_device):
    """`python bench.py --gpus 2` with no launcher around it (the way the driver runs it): with RCCL it must refuse loudly
    on a box that shows fewer than 2 devices -- never an N = 1 line under `--gpus 2` -- and with LR_DIST_BACKEND=gloo (ranks
    sharing the device) it starts the two ranks itself and the line proves the collective ran over both."""
    import subprocess
    import sys
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gaussians", "100000",
            "--views", "4", "--resolution", "640x360", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LR_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "needs 2 HIP devices" in r.stderr, (r.returncode, r.stderr[-2000:])
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]              # no bench line at all
    r = subprocess.run(args, env=dict(env, LR_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["dist_world_size"] == 2 and line["config"]["dist_backend"] == "gloo"
    chk = line["config"]["collective_check"]
    assert chk["all_reduce_of_ones"] == 2.0 and chk["ranks"] == 2


def test_bench_refuses_a_world_size_that_is_not_gpus(hip_device):
    """A launcher that starts 2 ranks for `--gpus 4` is an error, not a 2-rank line labelled otherwise."""
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--gaussians", "100000",
                      "--views", "4", "--resolution", "640x360", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extras"], 29629)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_bench_split_sharded_adam_reports_the_overlap_window(hip_device):
    """--exchange split-sharded-adam: the training step with the parameter all-gather in two parts (parallel.SplitShardedAdam);
    the line says how many bytes are joined inside the optimizer step (geometry) and how many are left in flight (SH)."""
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "100000",
                      "--views", "4", "--resolution", "640x360", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extras",
                      "--exchange", "split-sharded-adam"], 29631)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    cfg = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["config"]
    d = cfg["exchange_detail"]
    assert cfg["exchange"] == "split-sharded-adam" and cfg["collective_check"]["all_reduce_of_ones"] == 2.0
    assert 0 < d["all_gather_geometry_joined_in_step"] < d["all_gather_appearance_left_in_flight"]
    assert d["reduce_scatter"] == d["all_gather_geometry_joined_in_step"] + d["all_gather_appearance_left_in_flight"]
