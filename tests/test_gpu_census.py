"""Near-tie census against the reference's own arithmetic at scale: the dense formulation of
vqvae/ema_vector_quantizer.py:42-53 (sgemm + torch.argmin) on the same rows.  Rows on which this
library's indices differ are counted, and every one of them must be a near-tie in float64
(tools/near_tie_census.py); histograms may then differ by those rows only."""
import pytest
import torch

from tools.near_tie_census import census, dense_indices

pytestmark = pytest.mark.gpu


def _latents(dev, B, D, H, seed):
    return torch.randn(B, D, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))


@pytest.mark.parametrize("K,D,B,T", [(8192, 256, 256, 1.0), (8192, 256, 64, 2.5), (512, 256, 256, 1.0),
                                     (64, 256, 256, 2.5), (512, 64, 256, 1.0), (1000, 48, 64, 1.0)])
def test_census_against_the_dense_formulation_on_the_gpu(K, D, B, T):
    """rocBLAS sgemm as the reference's matmul: 256 Ki rows at the headline codebook, the reference's own
    codebook shapes (train_vqvae.py:147-148, improved_vq_config.py:8-17) and one odd channel count."""
    from vq_vae_amd import ops
    dev = torch.device("cuda:0")
    z = _latents(dev, B, D, 32, 11)
    E = torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(12))
    got = ops.search(z, E, ops.Workspace(dev), temperature=T, want_quantized=False)["indices"]
    c = census(z, E, got, dense_indices(z, E, T), T)
    print(f"K={K} D={D} T={T}: {c}")
    assert c["rows"] == B * 32 * 32 >= 65536
    assert c["not_near_ties"] == 0, c
    assert c["mismatches"] <= c["rows"] // 1000, c       # near-ties are rare events, not a systematic disagreement


def test_census_against_the_dense_formulation_on_the_host():
    """MKL sgemm on the host cores as the reference's matmul (its CPU path): 64 Ki rows, K=8192, D=256."""
    from vq_vae_amd import ops
    dev = torch.device("cuda:0")
    z = _latents(dev, 64, 256, 32, 21)
    E = torch.randn(8192, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(22))
    got = ops.search(z, E, ops.Workspace(dev), want_quantized=False)["indices"]
    torch.set_num_threads(16)
    c = census(z.cpu(), E.cpu(), got.cpu(), dense_indices(z.cpu(), E.cpu(), 1.0, chunk=16384))
    print(c)
    assert c["rows"] == 65536 and c["not_near_ties"] == 0, c


def test_census_after_training_steps():
    """The steady state of training: the codebook is the EMA of cluster means (codes close to the data,
    neighbours close to each other), where near-ties are likeliest."""
    from vq_vae_amd import ops
    dev = torch.device("cuda:0")
    K, D = 8192, 256
    z = _latents(dev, 128, D, 32, 31)
    E = torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(32))
    w = torch.zeros(K, D, device=dev); cs = torch.zeros(K, device=dev); u = torch.zeros(K, device=dev)
    ws = ops.Workspace(dev)
    for _ in range(3):
        ops.quantize_step(z, E, w, cs, u, beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=True, ws=ws)
    z2 = _latents(dev, 128, D, 32, 33)
    got = ops.search(z2, E, ws, want_quantized=False)["indices"]
    c = census(z2, E, got, dense_indices(z2, E))
    print(c)
    assert c["not_near_ties"] == 0, c
