#!/usr/bin/env python3
"""Times the EMA embedding-sum kernels at the benchmark size (K=8192, D=256, 4 Mi latents): NCHW,
row-major by counting sort, row-major by code ranges; uniform and skewed assignments.  Each form twice: called on
its own (the launch then starts with a max|z| reduction over the latents, vq_absmax_kernel) and as inside a
training step, where the search has left max|z| in the control block already (zmax_valid).

    python tools/scatter_bench.py [--rows N]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--K", type=int, default=8192)
    ap.add_argument("--D", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K, D, N = a.K, a.D, a.rows
    g = torch.Generator(device=dev).manual_seed(0)
    z_rows = torch.randn(N, D, device=dev, generator=g)
    z_nchw = z_rows.view(N // 1024, 32, 32, D).permute(0, 3, 1, 2).contiguous()
    cases = {"uniform": torch.randint(0, K, (N,), device=dev, generator=g, dtype=torch.int32),
             "one code": torch.full((N,), 7, device=dev, dtype=torch.int32),
             "zipf-like": (torch.rand(N, device=dev, generator=g).pow(8) * K).to(torch.int32).clamp_(0, K - 1)}
    ws = ops.Workspace(dev)
    gb = N * D * 4 / 1e9
    for name, idx in cases.items():
        t_sorted = timed(lambda: ops.embed_sums(z_rows, idx, K, ws))
        t_ranges = timed(lambda: ops.embed_sums(z_rows, idx, K, ws, sorted_rows=False))
        t_nchw = timed(lambda: ops.embed_sums(z_nchw, idx, K, ws))
        # the calls above left max|z| of these very latents in the control block: what a step's scatter finds
        s_sorted = timed(lambda: ops.embed_sums(z_rows, idx, K, ws, zmax_valid=True))
        s_nchw = timed(lambda: ops.embed_sums(z_nchw, idx, K, ws, zmax_valid=True))
        a_ = ops.embed_sums(z_rows, idx, K, ws)[0][0].clone()
        b_ = ops.reduce_slabs(ops.embed_sums(z_nchw, idx, K, ws)[0])
        err = float((a_ - b_).abs().max() / b_.abs().max())
        print(f"{name:10s} rows sorted {t_sorted:7.3f} ms ({gb / t_sorted:5.2f} TB/s)   rows by code range {t_ranges:7.3f} ms   "
              f"NCHW {t_nchw:7.3f} ms ({gb / t_nchw:5.2f} TB/s)   sorted vs NCHW rel diff {err:.1e}\n"
              f"{'':10s} in a step (max|z| known): rows sorted {s_sorted:7.3f} ms ({gb / s_sorted:5.2f} TB/s)   "
              f"NCHW {s_nchw:7.3f} ms ({gb / s_nchw:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
