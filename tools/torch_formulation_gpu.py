#!/usr/bin/env python3
"""How fast is the reference's own formulation on the same GPU?

The reference computes the step with dense torch operations: an N x K distance matrix, argmin, an
N x K one-hot matrix, and two more dense products for the gather and the EMA sums
(vqvae/ema_vector_quantizer.py:35-106).  This script writes that formulation out with stock torch
ops (rocBLAS sgemm and elementwise kernels), runs it on cuda:0 at batch sizes whose N x K matrices
fit comfortably, and prints latent vectors per second next to this library's step on the same
inputs.  It is a yardstick, not part of the product or of the tests.

    python tools/torch_formulation_gpu.py [rows ...]        (default: 65536 262144 1048576)

For every size it also prints the near-tie census: the rows on which this library's indices differ
from the dense formulation's (rocBLAS accumulates z.e in its own order), each classified in float64 as
a near-tie or not (tools/near_tie_census.py), for T = 1 and T = 2.5.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dense_step(inputs, E, ema_w, cs, usage, beta, decay, eps):
    """One training forward, op for op as the cited lines (fp32, no autograd)."""
    K, D = E.shape
    flat = inputs.permute(0, 2, 3, 1).contiguous().view(-1, D)                     # :35-36
    dist = (flat ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * flat @ E.t()  # :42-46
    idx = torch.argmin(dist, dim=1, keepdim=True)                                  # :53
    enc = torch.zeros(idx.shape[0], K, device=inputs.device)
    enc.scatter_(1, idx, 1)                                                        # :54-55
    q = (enc @ E).view(inputs.shape[0], inputs.shape[2], inputs.shape[3], D)       # :58
    counts = enc.sum(0)                                                            # :65
    usage += counts                                                                # :67
    cs.mul_(decay).add_(counts, alpha=1 - decay)                                   # :71-72
    n = cs.sum()                                                                   # :75
    cs.copy_((cs + eps) / (n + K * eps) * n)                                       # :76-80
    dw = enc.t() @ flat                                                            # :84
    ema_w.mul_(decay).add_(dw, alpha=1 - decay)                                    # :85
    E.copy_(ema_w / cs.unsqueeze(1))                                               # :89
    xin = inputs.permute(0, 2, 3, 1)
    loss = beta * torch.mean((q - xin) ** 2)                                       # :92
    q = (xin + (q - xin)).permute(0, 3, 1, 2).contiguous()                         # :95-98
    p = enc.mean(0)                                                                # :105
    perp = torch.exp(-(p * torch.log(p + 1e-10)).sum())                            # :106
    return q, loss, perp, idx


def main():
    from vq_vae_amd import ops
    rows = [int(a) for a in sys.argv[1:]] or [65536, 262144, 1048576]
    dev = torch.device("cuda:0")
    K, D = 8192, 256
    g = torch.Generator(device=dev).manual_seed(0)
    E0 = torch.randn(K, D, device=dev, generator=g)
    ws = ops.Workspace(dev)
    for N in rows:
        z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g)

        def fresh():
            return E0.clone(), torch.zeros(K, D, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)

        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        E, w, c, u = fresh()
        with torch.no_grad():
            t_ref = timed(lambda: dense_step(z, E, w, c, u, 3.0, 0.95, 1e-5), 3)
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        out = {}
        for mode in ("fp32", "filtered"):
            E, w, c, u = fresh()
            out[mode] = timed(lambda: ops.quantize_step(z, E, w, c, u, beta=3.0, decay=0.95, epsilon=1e-5,
                                                        temperature=1.0, training=True, ws=ws,
                                                        filtered=mode == "filtered"), 5)
        import json
        from tools.near_tie_census import census, dense_indices
        for T in (1.0, 2.5):
            got = ops.search(z, E0, ws, temperature=T, want_quantized=False)["indices"]
            c = census(z, E0, got, dense_indices(z, E0, T), T)
            c.update({"K": K, "D": D, "temperature": T})
            print("census " + json.dumps(c), flush=True)
        print(f"N={N}: dense torch formulation {t_ref * 1e3:.1f} ms = {N / t_ref / 1e6:.2f} M vec/s "
              f"(peak {peak:.0f} GiB) | this library fp32 {out['fp32'] * 1e3:.2f} ms = {N / out['fp32'] / 1e6:.1f} M vec/s, "
              f"filtered {out['filtered'] * 1e3:.2f} ms = {N / out['filtered'] / 1e6:.1f} M vec/s", flush=True)
        del z
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
