#!/usr/bin/env python3
"""Randomised comparison of the two 1x1-convolution kernels against float64 products: random batch shapes,
channel counts (inside and outside the kernels' blocks), both memory layouts, with and without bias.
vq_conv1x1 is checked for every shape it supports, vq_conv1x1_wgrad likewise.  Not collected by pytest:

    python tests/fuzz_conv.py [trials] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vq_vae_amd import ops  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.Generator(np.random.PCG64(seed))
    dev = torch.device("cuda:0")
    ws = ops.Workspace(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    bad = n_fwd = n_wg = 0
    for t in range(trials):
        B = int(rng.integers(1, 6))
        H = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 32]))
        W = int(rng.choice([1, 2, 4, 8, 9, 16, 32]))
        C = int(rng.choice([16, 32, 64, 128, 256, 4, 12, 36, 40, 272, 520]))
        O = int(rng.choice([1, 3, 4, 31, 32, 33, 64, 96, 100, 128, 256, 257, 320, 512]))
        cl = bool(rng.integers(0, 2))
        if t % 8 == 7:
            # a batch large enough for the weights-stationary kernel (32 Ki rows and more, whole
            # 256-channel output blocks); channels_last with H * W odd gives ragged 32-row tiles
            C = int(rng.choice([64, 128, 256]))
            O = int(rng.choice([256, 512]))
            H, W = (int(rng.choice([16, 32])), 32) if not cl else (int(rng.choice([31, 32, 33])), int(rng.choice([31, 32])))
            B = -(-32768 // (H * W)) + int(rng.integers(0, 6))
        scale = float(rng.choice([1.0, 1e-3, 30.0]))
        x = torch.randn(B, C, H, W, device=dev, generator=g) * scale
        gy = torch.randn(B, O, H, W, device=dev, generator=g)
        Wt = torch.randn(O, C, device=dev, generator=g) * 0.2
        bias = torch.randn(O, device=dev, generator=g) if rng.integers(0, 2) else None
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            gy = gy.contiguous(memory_format=torch.channels_last)
        X = x.permute(0, 2, 3, 1).reshape(-1, C).double()
        G = gy.permute(0, 2, 3, 1).reshape(-1, O).double()
        n = X.shape[0]
        tag = f"trial {t}: B={B} C={C} O={O} H={H} W={W} channels_last={cl} scale={scale}"
        if ops.conv1x1_supported(x, O, C):
            n_fwd += 1
            out = ops.conv1x1(x, Wt, bias, ws)
            want = X @ Wt.double().t() + (0 if bias is None else bias.double())
            got = out.permute(0, 2, 3, 1).reshape(-1, O).double()
            err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
            if not err <= 4e-6 or out.shape != (B, O, H, W):   # one fp32 chain over C products against float64 (C * 2^-24 is the worst case; 2.0e-6 was seen once in 1 500 shapes)
                bad += 1
                print(f"FAIL conv1x1 {tag}: rel err {err:.3e}", flush=True)
        if ops.conv1x1_wgrad_supported(gy, x):
            n_wg += 1
            dW, db = ops.conv1x1_wgrad(gy, x, ws)
            wantW, wantb = G.t() @ X, G.sum(0)
            tol = 4e-6 * n ** 0.5 + 1e-5
            eW = (dW.double() - wantW).abs().max().item() / max(1.0, wantW.abs().max().item() / n ** 0.5) / max(scale, 1e-30)
            eb = (db.double() - wantb).abs().max().item() / max(1.0, wantb.abs().max().item() / n ** 0.5)
            if not (eW <= tol and eb <= tol):
                bad += 1
                print(f"FAIL wgrad {tag}: dW err {eW:.3e} db err {eb:.3e} (tol {tol:.3e})", flush=True)
        if (t + 1) % 50 == 0:
            print(f"{t + 1} trials, {bad} failures", flush=True)
    print(f"fuzz_conv: {trials} problems ({n_fwd} through vq_conv1x1, {n_wg} through vq_conv1x1_wgrad), {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
