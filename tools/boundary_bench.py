#!/usr/bin/env python3
"""Side timings at the benchmark size (K=8192, D=256, 4 Mi latents): the training step on channels_last
against NCHW inputs, and the segment quant_conv -> quantizer -> post_quant_conv (QuantizerBoundary)
against the same segment composed from torch convolutions around the quantizer module."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import EMAVectorQuantizer, QuantizerBoundary, ops  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
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
    K, D = 8192, 256
    g = torch.Generator(device=dev).manual_seed(0)
    z = torch.randn(rows // 1024, D, 32, 32, device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g)
    zcl = z.contiguous(memory_format=torch.channels_last)
    for filt in (False, True):
        for name, zz in (("NCHW", z), ("channels_last", zcl)):
            ws = ops.Workspace(dev)
            st = [E.clone(), torch.zeros(K, D, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)]
            t = timed(lambda: ops.quantize_step(zz, st[0], st[1], st[2], st[3], beta=3.0, decay=0.95, epsilon=1e-5,
                                                temperature=1.0, training=True, ws=ws, filtered=filt))
            print(f"training step, {'filtered' if filt else 'fp32'} search, {name}: {t:.2f} ms", flush=True)
    del zcl
    wt = torch.randn(rows // 1024, D, 32, 32, device=dev, generator=g)
    for mode in ("filtered", "fp32"):
        b = QuantizerBoundary(D, K, D).to(dev)
        with torch.no_grad():
            b.quantize.embedding.weight.copy_(E)
        b.quantize.search_mode = mode
        b.train()
        hh = z.clone().requires_grad_(True)

        def fused(backward):
            out, loss, _ = b(hh)
            if backward:
                ((out * wt).sum() + loss).backward()
                hh.grad = None
                b.zero_grad(set_to_none=True)

        def unfused(backward):
            qq, loss, _ = b.quantize(b.quant_conv(hh))
            out = b.post_quant_conv(qq)
            if backward:
                ((out * wt).sum() + loss).backward()
                hh.grad = None
                b.zero_grad(set_to_none=True)

        with torch.no_grad():
            tf, tu = timed(lambda: fused(False), 3, 1), timed(lambda: unfused(False), 3, 1)
        tfb, tub = timed(lambda: fused(True), 3, 1), timed(lambda: unfused(True), 3, 1)
        print(f"quant_conv -> quantizer -> post_quant_conv, {mode}: forward {tf:.1f} ms (torch convolutions around the "
              f"module: {tu:.1f}), forward+backward {tfb:.1f} ms ({tub:.1f})", flush=True)
        del b, hh


if __name__ == "__main__":
    main()
