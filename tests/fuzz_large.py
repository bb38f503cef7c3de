#!/usr/bin/env python3
"""Randomised comparison of the large-batch search paths (the persistent kernel, whole-codebook sweeps with the
deferred temperature, the rows handed to the exact kernel's two lists) against the exact scalar kernel: random
codebook sizes, channel counts, batch sizes of 64 Ki to 320 Ki rows, both layouts, temperatures, value scales,
codebooks with near-duplicate rows, a few rows that are not finite.  Indices, distances (quotients) and outputs
must be equal bit for bit, the loss to 1e-6.  Prints one line per failure and a summary.

    python tests/fuzz_large.py [trials] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vq_vae_amd import _lib, ops  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.Generator(np.random.PCG64(seed))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    lib = _lib.load()
    ws = ops.Workspace(dev)
    bad, persistent, close_rows, listed_rows = 0, 0, 0, 0
    for t in range(trials):
        D = int(rng.choice([16, 32, 64, 128, 256, 48, 200]))
        K = int(rng.choice([64, 100, 512, 1000, 1024, 2048, 4096]))
        images = int(rng.choice([64, 128, 130, 256, 320]))
        hw = int(rng.choice([32, 16]))            # 32 x 32 or 16 x 16 latents per image
        if hw == 16:
            images *= 4
        layout = str(rng.choice(["nchw", "rows"]))
        T = float(rng.choice([1.0, 2.5, 0.7, 0.013, 60.0]))
        scale = float(10.0 ** rng.uniform(-2, 2))
        kind = str(rng.choice(["gauss", "neardup", "pairs", "nonfinite"]))
        E = torch.randn(K, D, device=dev, generator=g) * scale
        z = torch.randn(images, D, hw, hw, device=dev, generator=g) * scale
        if kind == "neardup":
            half = K // 2
            E[half: 2 * half] = E[:half] * (1 + torch.randint(-3, 4, (half, 1), device=dev, generator=g) * 2.0 ** -23)
        elif kind == "pairs":
            for p in range(max(1, K // 1024)):
                E[3 + 5 * p] = E[K // 2 + 7 * p + 1] * (1 + torch.randint(-2, 3, (D,), device=dev, generator=g) * 2.0 ** -23)
        elif kind == "nonfinite":
            z[1, 3, 2, 5] = float("nan"); z[2, 0, 0, 0] = float("inf"); z[3, :, 4, 4] = 3e19
        if layout == "rows":
            z = z.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
        N = z.numel() // D
        persistent += int(lib.vq_search_uses_persistent_kernel(N, K, D, 1 if layout == "nchw" else 0, hw * hw))
        a = ops.search(z, E, ws, temperature=T, want_quantized=True, want_dmin=True)
        ctl = a["control"][:8].cpu().tolist()
        listed_rows += ctl[0]; close_rows += ctl[6]
        ia, da, qa = a["indices"].clone(), a["dmin"].clone(), a["quantized"].clone()
        la = a["loss_partial"][: a["n_partial"]].double().sum().item()
        b = ops.search(z, E, ws, temperature=T, want_quantized=True, want_dmin=True, exact=True)
        lb = b["loss_partial"][: b["n_partial"]].double().sum().item()
        torch.cuda.synchronize()
        same = (torch.equal(ia, b["indices"]) and torch.equal(da.view(torch.int32), b["dmin"].view(torch.int32))
                and torch.equal(qa.view(torch.int32), b["quantized"].view(torch.int32))
                and (abs(la - lb) <= 1e-6 * abs(lb) or (la != la and lb != lb)))
        if not same:
            bad += 1
            nd = int((ia != b["indices"]).sum())
            print(f"FAIL trial {t}: K={K} D={D} N={N} {layout} T={T} scale={scale:.3g} {kind}: {nd} indices differ, loss {la} vs {lb}", flush=True)
        del z, E
    print(f"{trials} problems, {bad} failures; {persistent} on the persistent kernel; rows listed for the exact kernel: "
          f"{listed_rows} not finite, {close_rows} close under a temperature")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
