"""fp32 search (with the fused gather) over codebook sizes and channel counts other than the benchmark's:
time, TFLOP/s and vectors/s at 1 Mi latents, and the filtered search where it applies.
python tools/shape_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import _lib, ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    ws = ops.Workspace(dev)
    N = 1 << 20
    g = torch.Generator(device=dev).manual_seed(0)
    print("K, D: fp32 search ms (TFLOP/s, M vectors/s) | filtered search ms")
    for K, D in ((64, 256), (512, 256), (2048, 256), (8192, 256), (512, 64), (8192, 64), (8192, 128), (1000, 48), (512, 100), (8192, 200)):
        z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g)
        E = torch.randn(K, D, device=dev, generator=g)
        ms = timed(lambda: ops.search(z, E, ws, want_quantized=True))
        line = f"{K:5d}, {D:3d}: {ms:8.3f} ms ({2.0 * N * K * D / ms / 1e9:6.1f} TFLOP/s, {N / ms / 1e3:7.1f} M vec/s)"
        if K <= 1024 and D in (16, 32, 64, 128, 256):
            # short sweeps run the persistent kernel; the one-workgroup-per-tile kernel on the same input for comparison
            lib = _lib.load()
            lib.vq_internal_tune(0, 0)
            line += f" [one-tile kernel {timed(lambda: ops.search(z, E, ws, want_quantized=True)):.3f} ms]"
            lib.vq_internal_tune(0, 1)
        if D in (64, 128, 256) and K >= 512:
            fm = timed(lambda: ops.search(z, E, ws, want_quantized=True, filtered=True))
            line += f" | {fm:7.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    main()
