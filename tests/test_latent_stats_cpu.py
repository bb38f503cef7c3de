"""The latent-space analyses of vq_vae_amd/latent_stats.py against SciPy / NumPy (float64, CPU).
The reference describes these analyses in README.md:21-34 and holds no code for them: parity unpinned."""
import numpy as np
import scipy.stats as st
import torch

from inputs import gauss
from vq_vae_amd.latent_stats import LatentMoments, latent_distribution_stats, quantization_consistency


def test_moments_match_scipy_over_batches():
    rng = np.random.default_rng(3)
    D = 12
    mix = rng.normal(size=(D, D)) * 0.4 + np.eye(D)
    x = (rng.standard_gamma(3.0, size=(5000, D)) - 3.0) @ mix + rng.normal(size=D)      # skewed, correlated, shifted
    acc = LatentMoments(D, chunk_rows=777)
    # three batches: two NCHW (one of them channels_last) and one (N, D)
    a = torch.from_numpy(x[:2048].reshape(2, 32, 32, D).transpose(0, 3, 1, 2).copy())
    b = torch.from_numpy(x[2048:4096].reshape(2, 32, 32, D).transpose(0, 3, 1, 2).copy()).contiguous(memory_format=torch.channels_last)
    c = torch.from_numpy(x[4096:].copy())
    for t in (a, b, c):
        acc.update(t)
    r = acc.result()
    assert r["count"] == 5000
    assert np.allclose(r["mean"].numpy(), x.mean(0), rtol=1e-10, atol=1e-12)
    assert np.allclose(r["variance"].numpy(), x.var(0), rtol=1e-9)
    assert np.allclose(r["skewness"].numpy(), st.skew(x, axis=0), rtol=1e-7, atol=1e-9)
    assert np.allclose(r["excess_kurtosis"].numpy(), st.kurtosis(x, axis=0), rtol=1e-7, atol=1e-9)
    assert np.allclose(r["correlation"].numpy(), np.corrcoef(x, rowvar=False), rtol=1e-8, atol=1e-10)
    jb = np.array([st.jarque_bera(x[:, j]).statistic for j in range(D)])
    assert np.allclose(r["jarque_bera"].numpy(), jb, rtol=1e-7)
    kl = 0.5 * (x.var(0) + x.mean(0) ** 2 - 1 - np.log(x.var(0)))
    assert np.allclose(r["kl_to_standard_normal"].numpy(), kl, rtol=1e-9)
    off = np.corrcoef(x, rowvar=False) - np.eye(D)
    assert abs(r["max_abs_correlation"] - np.abs(off).max()) < 1e-9
    assert abs(r["mean_abs_correlation"] - np.abs(off).sum() / (D * (D - 1))) < 1e-9


def test_standard_normal_latents_score_near_zero():
    z = torch.from_numpy(gauss(5, (8, 16, 32, 32)))
    r = latent_distribution_stats(z)
    s = r["summary"]
    assert abs(s["mean_of_means"]) < 0.02 and abs(s["mean_variance"] - 1) < 0.02
    assert s["mean_kl_to_standard_normal"] < 1e-3 and r["mean_abs_correlation"] < 0.03
    assert s["mean_abs_skewness"] < 0.08 and s["mean_abs_excess_kurtosis"] < 0.15


def test_quantization_consistency():
    z = torch.from_numpy(gauss(6, (2, 8, 4, 4)))
    same = quantization_consistency(z, z)
    assert same["relative_error"] == 0.0 and abs(same["mean_cosine_similarity"] - 1) < 1e-12
    q = 0.5 * z
    half = quantization_consistency(z, q)
    assert abs(half["relative_error"] - 0.25) < 1e-12 and abs(half["mean_relative_error_per_vector"] - 0.25) < 1e-12
    assert abs(half["sqnr_db"] - 10 * np.log10(4.0)) < 1e-9 and abs(half["mean_cosine_similarity"] - 1) < 1e-12
    zz = z.permute(0, 2, 3, 1).reshape(-1, 8).double().numpy()
    qn = torch.from_numpy(gauss(7, (2, 8, 4, 4)))
    got = quantization_consistency(z, qn.contiguous(memory_format=torch.channels_last))
    qq = qn.permute(0, 2, 3, 1).reshape(-1, 8).double().numpy()
    assert abs(got["relative_error"] - ((zz - qq) ** 2).sum() / (zz ** 2).sum()) < 1e-12
    cos = (zz * qq).sum(1) / np.sqrt((zz ** 2).sum(1) * (qq ** 2).sum(1))
    assert abs(got["mean_cosine_similarity"] - cos.mean()) < 1e-12
