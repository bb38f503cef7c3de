"""The persistent short-sweep search kernel (small codebooks, large batches): against the CPU oracle on
sampled rows, against the one-tile-per-workgroup kernel bit for bit, and on the cases its cyclic sweep
could get wrong -- ties across the point where a sweep wraps around the codebook's end, rows that are not
finite, a codebook that is not finite, ragged batches, temperatures.

The kernel only takes batches of at least 128 Ki rows, so the inputs here are larger than the oracle
sweeps in seconds: the oracle checks a sample of rows, the other kernel checks all of them."""
import numpy as np
import pytest
import torch

from oracle import vq_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    from vq_vae_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    yield lib, ops, dev
    lib.vq_internal_tune(0, 1)


def _both(lib, ops, dev, z, E, T=1.0):
    """(indices, distances, quantized, loss sum) from the persistent kernel and from the other one."""
    out = []
    for enable in (1, 0):
        lib.vq_internal_tune(0, enable)
        ws = ops.Workspace(dev)
        r = ops.search(z, E, ws, temperature=T, want_quantized=True, want_dmin=True)
        torch.cuda.synchronize()
        out.append((r["indices"].clone(), r["dmin"].clone(), r["quantized"].clone(),
                    float(r["loss_partial"][: r["n_partial"]].double().sum()), r["idx32"][: r["N"]].clone()))
    lib.vq_internal_tune(0, 1)
    return out


def _same(a, b):
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
    assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
    # (a batch with rows that are not finite has a loss that is not finite, in the reference too)
    assert (np.isnan(a[3]) and np.isnan(b[3])) or a[3] == b[3] or abs(a[3] - b[3]) <= 1e-6 * max(abs(b[3]), 1e-30)
    assert torch.equal(a[4], b[4]) and torch.equal(a[4].long(), a[0])


SHAPES = [
    # K, D, (B, H): NCHW (B, D, H, H); B = 0 means row-major with H rows
    (64, 256, (160, 32)),      # improved_vq_config.py:8-17, image resident in LDS
    (512, 256, (136, 32)),     # train_vqvae.py:147-148, stages cycling through LDS
    (512, 64, (576, 16)),      # benchmark configuration 1's codebook, resident, H*W = 256
    (100, 32, (2200, 8)),      # K not a multiple of 64, H*W = 64
    (1000, 128, (130, 32)),    # 16 stages, the last one ragged
    (320, 256, (0, 140001)),   # row-major, a batch that is not a multiple of 32 rows
    (64, 64, (0, 262144)),     # row-major, resident
]


@pytest.mark.parametrize("K,D,shape", SHAPES)
@pytest.mark.parametrize("T", [1.0, 2.5])
def test_short_sweep_equals_oracle_and_the_other_kernel(K, D, shape, T, env):
    lib, ops, dev = env
    B, H = shape
    g = torch.Generator(device=dev).manual_seed(K + D)
    z = torch.randn((B, D, H, H) if B else (H, D), device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g) * 0.8
    new, old = _both(lib, ops, dev, z, E, T)
    _same(new, old)
    # the oracle on 3000 rows spread over the batch (first, last and random tiles)
    rows = z.permute(0, 2, 3, 1).reshape(-1, D) if B else z
    N = rows.shape[0]
    pick = torch.cat([torch.arange(0, 64), torch.arange(N - 64, N),
                      torch.randint(0, N, (2872,), generator=torch.Generator().manual_seed(5))]).to(dev)
    o_idx, o_d = vo.search(rows[pick].cpu().numpy(), E.cpu().numpy(), T)
    assert np.array_equal(new[0][pick].cpu().numpy(), o_idx)
    assert np.array_equal(new[1][pick].cpu().numpy().view(np.uint32), o_d.view(np.uint32))
    # quantized rows are codebook rows passed through z + (e - z)
    q = new[2].permute(0, 2, 3, 1).reshape(-1, D) if B else new[2]
    want = rows[pick] + (E[new[0][pick]] - rows[pick])
    assert torch.equal(q[pick].view(torch.int32), want.view(torch.int32))


def test_ties_across_the_wrap_keep_the_lowest_index(env):
    """A sweep that starts in the middle of the codebook sees the high codes first.  With the second half
    of the codebook a copy of the first, every row has two exact winners, k and k + K/2, in different
    runs of the sweep for the waves that wrap: the lower one must be named, as torch.argmin does."""
    lib, ops, dev = env
    g = torch.Generator(device=dev).manual_seed(3)
    for K, D, B in ((512, 256, 136), (128, 256, 136), (512, 64, 600)):
        H = 32 if D == 256 else 16
        z = torch.randn(B, D, H, H, device=dev, generator=g)
        half = torch.randn(K // 2, D, device=dev, generator=g)
        E = torch.cat([half, half]).contiguous()
        new, old = _both(lib, ops, dev, z, E)
        _same(new, old)
        assert int(new[0].max()) < K // 2
        # and a codebook of identical rows: index 0 everywhere
        E1 = half[:1].repeat(K, 1).contiguous()
        new, old = _both(lib, ops, dev, z, E1)
        _same(new, old)
        assert int(new[0].max()) == 0


def test_rows_that_are_not_finite_and_a_codebook_that_is_not(env):
    lib, ops, dev = env
    g = torch.Generator(device=dev).manual_seed(4)
    K, D = 256, 128
    z = torch.randn(160, D, 32, 32, device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g)
    zb = z.clone()
    zb[0, 3, 0, 5] = float("nan")          # row 5
    zb[7, 0, 31, 31] = float("inf")        # last row of image 7
    zb[159, 100, 16, :] = -float("inf")    # 32 consecutive rows = one whole tile
    zb[80, :, 2, 2] = 3e18                 # finite, but its squared norm overflows
    new, old = _both(lib, ops, dev, zb, E)
    _same(new, old)
    rows = zb.permute(0, 2, 3, 1).reshape(-1, D)
    bad = torch.nonzero(~torch.isfinite(rows).all(1) | (rows.abs().max(1).values > 1e18)).reshape(-1)
    assert bad.numel() == 35
    o_idx, _ = vo.search(rows[bad].cpu().numpy(), E.cpu().numpy())
    assert np.array_equal(new[0][bad].cpu().numpy(), o_idx)
    # a NaN in the codebook: every row goes to the exact kernel, which follows torch.argmin's NaN rule
    Eb = E.clone()
    Eb[77, 5] = float("nan")
    new, old = _both(lib, ops, dev, z, Eb)
    _same(new, old)
    assert bool((new[0] == 77).all())


def test_training_step_through_the_module_uses_the_kernel(env):
    """One training forward + backward of the module at a shape the persistent kernel takes, against the
    same step with the kernel switched off: state, outputs and gradient bit for bit, loss to 1e-6."""
    lib, ops, dev = env
    from vq_vae_amd import EMAVectorQuantizer
    g = torch.Generator(device=dev).manual_seed(6)
    K, D = 512, 256
    z0 = torch.randn(136, D, 32, 32, device=dev, generator=g)
    E0 = torch.randn(K, D, device=dev, generator=g)
    res = []
    for enable in (1, 0):
        lib.vq_internal_tune(0, enable)
        m = EMAVectorQuantizer(K, D, 3.0, decay=0.95).to(dev)
        with torch.no_grad():
            m.embedding.weight.copy_(E0)
        m.train()
        z = z0.clone().requires_grad_(True)
        q, loss, (perp, idx, _) = m(z, temperature=2.5)
        (q.square().sum() * 1e-3 + loss).backward()
        torch.cuda.synchronize()
        res.append((q.detach(), float(loss.detach()), float(perp.detach()), idx, z.grad, m.embedding.weight.detach().clone(),
                    m.ema_w.clone(), m.ema_cluster_size.clone(), m.usage_count.clone()))
    lib.vq_internal_tune(0, 1)
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1]) and a[2] == b[2]
    assert torch.allclose(a[4], b[4], rtol=1e-6, atol=0)      # the gradient carries the loss's scalar factor
    for i in (5, 6, 7, 8):
        assert torch.equal(a[i], b[i])
