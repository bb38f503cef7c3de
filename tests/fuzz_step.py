#!/usr/bin/env python3
"""Randomised comparison of the whole training step (fp32 search) against the CPU oracle: random codebook
sizes, channel counts (inside and outside the matrix-core set), batch shapes, layouts (NCHW, channels_last,
rows), temperatures and value scales; two steps per problem.  Prints one line per failure and a summary.

    python tests/fuzz_step.py [trials] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vq_oracle as vo  # noqa: E402
from vq_vae_amd import ops  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.Generator(np.random.PCG64(seed))
    dev = torch.device("cuda:0")
    bad = 0
    for t in range(trials):
        D = int(rng.choice([4, 8, 16, 24, 32, 64, 100, 128, 256]))
        K = int(rng.choice([1, 2, 7, 64, 65, 100, 512, 1000, 4096]))
        B, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 20)), int(rng.integers(1, 20))
        layout = str(rng.choice(["nchw", "channels_last", "rows"]))
        T = float(rng.choice([1.0, 1.0, 2.5, 0.7]))
        scale = float(10.0 ** rng.uniform(-3, 3))
        decay = float(rng.choice([0.95, 0.7, 0.99]))
        E0 = (rng.standard_normal((K, D)) * scale * 0.7).astype(np.float32)
        ws = ops.Workspace(dev)
        g = {"E": torch.from_numpy(E0).to(dev), "ema_w": torch.zeros(K, D, device=dev),
             "cs": torch.zeros(K, device=dev), "usage": torch.zeros(K, device=dev)}
        ok = True
        for step in range(2):
            z = (rng.standard_normal((B, D, H, W)) * scale).astype(np.float32)
            st = vo.OracleState(g["E"].cpu().numpy().copy(), g["cs"].cpu().numpy().copy(),
                                g["ema_w"].cpu().numpy().copy(), g["usage"].cpu().numpy().copy())
            ref = vo.forward(z, st, 3.0, decay, 1e-5, T, True)
            zt = torch.from_numpy(z).to(dev)
            if layout == "channels_last":
                zt = zt.contiguous(memory_format=torch.channels_last)
            elif layout == "rows":
                zt = zt.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
            out = ops.quantize_step(zt, g["E"], g["ema_w"], g["cs"], g["usage"], beta=3.0, decay=decay, epsilon=1e-5,
                                    temperature=T, training=True, ws=ws)
            torch.cuda.synchronize()
            q = out.quantized
            if layout == "rows":
                q = q.view(B, H, W, D).permute(0, 3, 1, 2)
            errs = {
                "idx": int((out.indices.cpu().numpy().reshape(-1) != ref["indices"].reshape(-1)).sum()),
                "q": 0 if np.array_equal(q.contiguous().cpu().numpy().view(np.uint32), ref["quantized"].view(np.uint32)) else 1,
                "loss": abs(out.loss.item() - ref["loss"]) > 1e-5 * abs(ref["loss"]),
                "perp": abs(out.perplexity.item() - ref["perplexity"]) > 1e-5 * abs(ref["perplexity"]),
                "E": rel(g["E"].cpu().numpy(), st.embedding) > 1e-5 if np.isfinite(st.embedding).all() else 0,
                "ema_w": rel(g["ema_w"].cpu().numpy(), st.ema_w) > 1e-5,
                "cs": rel(g["cs"].cpu().numpy(), st.ema_cluster_size) > 1e-5,
                "usage": not np.array_equal(g["usage"].cpu().numpy(), st.usage_count),
            }
            if any(errs.values()):
                ok = False
                print(f"FAIL trial {t} step {step}: K={K} D={D} shape=({B},{D},{H},{W}) {layout} T={T} scale={scale:.3g} decay={decay}: "
                      f"{ {k: v for k, v in errs.items() if v} }", flush=True)
                break
        bad += 0 if ok else 1
    print(f"{trials} problems, {bad} failed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
