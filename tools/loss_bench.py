#!/usr/bin/env python
"""Fused L1+DSSIM loss (forward + backward) vs the torch composition the reference uses (utils/loss.py), ms per call."""
import argparse, json, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luciddreamer_amd.loss import l1_dssim_loss     # noqa: E402


def torch_loss(img, gt, win, lam=0.2):
    C = img.shape[0]
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=C)[0]
    mu1, mu2 = conv(img), conv(gt)
    s1, s2, s12 = conv(img * img) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(img * gt) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - lam) * (img - gt).abs().mean() + lam * (1 - m.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resolution", default="1920x1080")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    W, H = (int(v) for v in a.resolution.split("x"))
    dev = torch.device("cuda:0")
    gt = torch.rand(3, H, W, device=dev)
    img = (0.7 * gt + 0.3 * torch.rand(3, H, W, device=dev)).requires_grad_(True)
    g = torch.tensor([__import__("math").exp(-(x - 5) ** 2 / 4.5) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    win = (g @ g.t()).expand(3, 1, 11, 11).contiguous().to(dev)
    res = {}
    for name, fn in (("fused", lambda: l1_dssim_loss(img, gt, 0.2)), ("torch", lambda: torch_loss(img, gt, win))):
        for _ in range(3):
            img.grad = None
            fn().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            img.grad = None
            fn().backward()
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / a.reps * 1e3, 4)
    n = 3 * H * W
    res["fused_GBps_algorithmic"] = round(n * 4 * (2 + 3 + 3 + 2 + 1) / (res["fused"] * 1e-3) / 1e9, 1)
    print(json.dumps({"workload": f"L1+DSSIM fwd+bwd, 3x{H}x{W}", "ms": res}))


if __name__ == "__main__":
    main()
