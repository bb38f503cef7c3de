"""Latent-space analyses the reference's README describes but holds no code for (README.md:21-24
"latent distribution analysis": statistics of the latent vectors, distance to a normal distribution,
correlation between dimensions; README.md:31-34: consistency before / after quantization).

These sit beside the hot path, not on it: a few reductions over the latents that the quantizer has
already read, written with stock torch operations so that they run on whatever device holds the
tensors.  The definitions are this package's own (parity unpinned: there is nothing in the reference to
compare with); the tests check them against SciPy / NumPy in float64.
"""
from __future__ import annotations

import math

import torch


def _rows(latents: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) in either memory format, or (N, C) -> (N, C) rows (a view where possible)."""
    if latents.dim() == 4:
        return latents.permute(0, 2, 3, 1).reshape(-1, latents.shape[1])
    if latents.dim() == 2:
        return latents
    raise RuntimeError(f"expected (B, C, H, W) or (N, D) latents, got {tuple(latents.shape)}")


class LatentMoments:
    """Accumulates, over batches of latents, the power sums up to order four per channel and the
    channel co-moment matrix (all in float64), then reports per-channel mean / variance / skewness /
    excess kurtosis, the distance of every channel's marginal from N(0, 1), and the correlation
    between channels."""

    def __init__(self, channels: int, device="cpu", chunk_rows: int = 1 << 18):
        self.D = int(channels)
        self.device = torch.device(device)
        self.chunk = int(chunk_rows)
        self.n = 0
        self.s = [torch.zeros(self.D, dtype=torch.float64, device=self.device) for _ in range(4)]
        self.xx = torch.zeros(self.D, self.D, dtype=torch.float64, device=self.device)

    def update(self, latents: torch.Tensor) -> None:
        rows = _rows(latents)
        if rows.shape[1] != self.D:
            raise RuntimeError(f"expected {self.D} channels, got {rows.shape[1]}")
        for lo in range(0, rows.shape[0], self.chunk):
            x = rows[lo: lo + self.chunk].to(device=self.device, dtype=torch.float64)
            x2 = x * x
            self.s[0] += x.sum(0)
            self.s[1] += x2.sum(0)
            self.s[2] += (x2 * x).sum(0)
            self.s[3] += (x2 * x2).sum(0)
            self.xx += x.t() @ x
            self.n += x.shape[0]

    def result(self) -> dict:
        if self.n == 0:
            raise RuntimeError("no latents accumulated")
        n = float(self.n)
        m1, m2, m3, m4 = (s / n for s in self.s)                    # raw moments
        var = (m2 - m1 * m1).clamp_min(0.0)                          # biased (population) variance
        std = var.sqrt()
        c3 = m3 - 3 * m1 * m2 + 2 * m1 ** 3                          # central moments
        c4 = m4 - 4 * m1 * m3 + 6 * m1 * m1 * m2 - 3 * m1 ** 4
        safe = var.clamp_min(1e-300)
        skew = c3 / safe ** 1.5
        kurt = c4 / (safe * safe) - 3.0
        # distance from the standard normal: KL( N(mean, var) || N(0, 1) ) per channel, and the
        # Jarque-Bera statistic n/6 (S^2 + K^2/4) for the shape of the marginal
        kl = 0.5 * (var + m1 * m1 - 1.0 - safe.log())
        jb = n / 6.0 * (skew * skew + kurt * kurt / 4.0)
        cov = self.xx / n - torch.outer(m1, m1)
        denom = torch.outer(std, std).clamp_min(1e-300)
        corr = cov / denom
        off = corr - torch.diag(torch.diag(corr))
        d = self.D
        mean_abs_off = float(off.abs().sum() / max(d * (d - 1), 1))
        return {
            "count": self.n,
            "mean": m1, "variance": var, "skewness": skew, "excess_kurtosis": kurt,
            "kl_to_standard_normal": kl, "jarque_bera": jb,
            "correlation": corr,
            "mean_abs_correlation": mean_abs_off,
            "max_abs_correlation": float(off.abs().max()) if d > 1 else 0.0,
            "summary": {
                "mean_of_means": float(m1.mean()), "mean_variance": float(var.mean()),
                "mean_kl_to_standard_normal": float(kl.mean()),
                "mean_abs_skewness": float(skew.abs().mean()), "mean_abs_excess_kurtosis": float(kurt.abs().mean()),
            },
        }


def latent_distribution_stats(latent_batches, channels: int = None, device=None) -> dict:
    """LatentMoments over an iterable of latent batches (or a single tensor)."""
    if isinstance(latent_batches, torch.Tensor):
        latent_batches = [latent_batches]
    acc = None
    for z in latent_batches:
        if acc is None:
            acc = LatentMoments(channels or _rows(z).shape[1], device if device is not None else z.device)
        acc.update(z)
    if acc is None:
        raise RuntimeError("no latents given")
    return acc.result()


def quantization_consistency(latents: torch.Tensor, quantized: torch.Tensor) -> dict:
    """How much the quantizer moves the latents (README.md:33, "consistency before / after
    quantization"): relative squared error ||z - q||^2 / ||z||^2 over the batch, its per-vector mean,
    the mean cosine similarity between a latent and its code, and the signal-to-quantization-noise
    ratio in dB.  ``quantized`` is the quantizer's first output for ``latents``."""
    z = _rows(latents).double()
    q = _rows(quantized.detach()).to(z.device).double()
    if z.shape != q.shape:
        raise RuntimeError(f"shapes differ: {tuple(z.shape)} vs {tuple(q.shape)}")
    err = ((z - q) ** 2).sum(1)
    zn = (z * z).sum(1)
    qn = (q * q).sum(1)
    rel_total = float(err.sum() / zn.sum().clamp_min(1e-300))
    cos = (z * q).sum(1) / (zn.sqrt() * qn.sqrt()).clamp_min(1e-300)
    return {
        "relative_error": rel_total,
        "mean_relative_error_per_vector": float((err / zn.clamp_min(1e-300)).mean()),
        "mean_cosine_similarity": float(cos.mean()),
        "sqnr_db": float(-10.0 * math.log10(max(rel_total, 1e-300))),
    }


@torch.no_grad()
def interpolation_smoothness(codebook: torch.Tensor, start: torch.Tensor, end: torch.Tensor, steps: int = 16,
                             ws=None) -> dict:
    """Smoothness of quantized interpolation paths (the reference's README.md:31-32 lists
    "interpolation smoothness" among its latent-space continuity analyses; it holds no code for it,
    so the definition is this package's own).

    For P pairs of latent vectors ``start[p]``, ``end[p]`` ((P, D) tensors on the GPU) the points
    z_t = (1 - t) start + t end, t = 0, 1/steps, ..., 1, are quantized with the HIP search
    (nearest code, lowest index on ties, as vqvae/ema_vector_quantizer.py:42-53).  Reported, each as
    the mean over the pairs plus the per-pair tensors:

    * ``code_switches``      how often the code changes along a path (0 .. steps)
    * ``distinct_codes``     how many different codes a path visits
    * ``path_length_ratio``  sum of ||q_{t+1} - q_t|| over ||q_last - q_first|| (1 = the quantized path is
                             as direct as its end points allow; NaN where both ends share a code)
    * ``max_jump_ratio``     largest single step ||q_{t+1} - q_t|| over ||end - start|| / steps, the
                             length of one interpolation step before quantization
    * ``quantization_error`` mean ||z_t - q_t||^2 along the path
    """
    from . import ops
    if start.shape != end.shape or start.dim() != 2:
        raise RuntimeError(f"start and end must both be (P, D), got {tuple(start.shape)} and {tuple(end.shape)}")
    if steps < 1:
        raise ValueError("steps must be at least 1")
    E = ops._f32c(codebook, "codebook")
    a = ops._f32c(start, "start")
    b = ops._f32c(end, "end")
    P, D = a.shape
    t = torch.linspace(0.0, 1.0, steps + 1, device=a.device, dtype=torch.float32).view(1, steps + 1, 1)
    z = (a.unsqueeze(1) * (1.0 - t) + b.unsqueeze(1) * t).reshape(P * (steps + 1), D).contiguous()
    ws = ws or ops.Workspace(a.device)
    idx = ops.search(z, E, ws, want_quantized=False)["indices"].view(P, steps + 1)
    q = E[idx]                                                   # (P, steps + 1, D)
    step_len = (q[:, 1:] - q[:, :-1]).double().norm(dim=2)        # (P, steps)
    direct = (q[:, -1] - q[:, 0]).double().norm(dim=1)
    switches = (idx[:, 1:] != idx[:, :-1]).sum(1)
    srt = idx.sort(dim=1).values
    distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
    nan = torch.full_like(direct, float("nan"))
    ratio = torch.where(direct > 0, step_len.sum(1) / direct.clamp_min(1e-300), nan)
    unit = (b - a).double().norm(dim=1) / steps
    jump = torch.where(unit > 0, step_len.max(1).values / unit.clamp_min(1e-300), torch.full_like(unit, float("nan")))
    qerr = (z.view(P, steps + 1, D) - q).double().pow(2).sum(2).mean(1)

    def mean(x):
        x = x.double()
        ok = ~torch.isnan(x)
        return float(x[ok].mean()) if bool(ok.any()) else float("nan")

    return {"steps": steps, "pairs": P, "indices": idx,
            "code_switches": mean(switches), "distinct_codes": mean(distinct),
            "path_length_ratio": mean(ratio), "max_jump_ratio": mean(jump), "quantization_error": mean(qerr),
            "per_pair": {"code_switches": switches, "distinct_codes": distinct, "path_length_ratio": ratio,
                         "max_jump_ratio": jump, "quantization_error": qerr}}
