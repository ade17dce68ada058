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


def test_bench_runs_under_two_ranks_and_reports_the_world_size(hip_device):
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gaussians", "100000",
                      "--views", "4", "--resolution", "640x360", "--no-cpu-baseline"], 29613)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["dist_world_size"] == 2 and line["config"]["dist_backend"] == "gloo"
    assert line["scaling"] == "weak" and line["value"] > 0 and line["entry_points"] is None
