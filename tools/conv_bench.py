#!/usr/bin/env python3
"""Times vq_conv1x1 against torch's (MIOpen / rocBLAS) 1x1 convolution at the benchmark size:
256 -> 256 channels over 4 Mi latent rows, NCHW and channels_last."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
    C = 256
    x = torch.randn(rows // 1024, C, 32, 32, device=dev)
    W = torch.randn(C, C, device=dev) * 0.06
    b = torch.randn(C, device=dev)
    ws = ops.Workspace(dev)
    flop = 2.0 * rows * C * C
    base = None
    for name, xx in (("NCHW", x), ("channels_last", x.contiguous(memory_format=torch.channels_last))):
        t_k = timed(lambda: ops.conv1x1(xx, W, b, ws))
        t_t = timed(lambda: torch.nn.functional.conv2d(xx, W[:, :, None, None], b))
        a = ops.conv1x1(xx, W, b, ws)
        r = torch.nn.functional.conv2d(xx, W[:, :, None, None], b)
        err = float((a - r).abs().max() / r.abs().max())
        if base is None:
            base = (a.clone(), r.clone())
        else:
            print(f"  channels_last vs NCHW: vq_conv1x1 bit-equal {torch.equal(a.contiguous(), base[0])}, torch conv2d rel diff "
                  f"{float((r.contiguous() - base[1]).abs().max() / base[1].abs().max()):.1e}")
        print(f"{name:14s} vq_conv1x1 {t_k:6.2f} ms ({flop / t_k / 1e9:5.1f} TFLOP/s)   torch conv2d {t_t:6.2f} ms "
              f"({flop / t_t / 1e9:5.1f} TFLOP/s)   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
