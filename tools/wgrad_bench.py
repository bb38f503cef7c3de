"""Times vq_conv1x1_wgrad against torch's convolution weight gradient at the headline batch
(4 Mi rows, 256 -> 256 channels), both layouts.  Run on the GPU box: python tools/wgrad_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    dev = torch.device("cuda:0")
    B, C, H, W = 4096, 256, 32, 32
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    gy = torch.randn(B, C, H, W, device=dev, generator=g)
    ws = ops.Workspace(dev)
    flops = 2.0 * B * H * W * C * C
    for name, xx, gg in (("nchw", x, gy), ("channels_last", x.contiguous(memory_format=torch.channels_last),
                                          gy.contiguous(memory_format=torch.channels_last))):
        ms = timed(lambda: ops.conv1x1_wgrad(gg, xx, ws))
        ref = timed(lambda: (torch.nn.grad.conv2d_weight(xx, (C, C, 1, 1), gg), gg.sum((0, 2, 3))), reps=5)
        dW, db = ops.conv1x1_wgrad(gg, xx, ws)
        tW = torch.nn.grad.conv2d_weight(xx, (C, C, 1, 1), gg).reshape(C, C)
        # fp64 sums of the first four output channels, image by image
        want = torch.zeros(4, C, dtype=torch.float64, device=dev)
        for b0 in range(0, B, 256):
            X = xx[b0:b0 + 256].permute(0, 2, 3, 1).reshape(-1, C).double()
            G = gg[b0:b0 + 256, :4].permute(0, 2, 3, 1).reshape(-1, 4).double()
            want += G.t() @ X
        scale = want.abs().max().item()
        err = (dW[:4].double() - want).abs().max().item() / scale
        terr = (tW[:4].double() - want).abs().max().item() / scale
        print(f"{name}: vq_conv1x1_wgrad {ms:.3f} ms ({flops / ms / 1e9:.1f} TFLOP/s, "
              f"{2 * x.numel() * 4 / ms / 1e6:.0f} GB/s)  torch conv2d_weight + bias sum {ref:.3f} ms  "
              f"error against fp64 / max|dW|: kernel {err:.2e}, torch {terr:.2e}", flush=True)


if __name__ == "__main__":
    main()
