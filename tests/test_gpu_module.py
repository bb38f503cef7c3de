"""GPU tests of the drop-in EMAVectorQuantizer module against the reference's golden outputs."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from inputs import gauss
from oracle import vq_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make(K, D, beta, decay, E, dev):
    from vq_vae_amd import EMAVectorQuantizer
    m = EMAVectorQuantizer(num_embeddings=K, embedding_dim=D, commitment_cost=beta, decay=decay).to(dev)
    with torch.no_grad():
        m.embedding.weight.copy_(torch.from_numpy(E))
    return m


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_*.npz"))))
def test_forward_matches_reference_outputs(path, dev):
    g = np.load(path)
    K, D = g["E"].shape
    z = gauss(int(g["z_seed"]), g["quantized"].shape)
    m = make(K, D, float(g["beta"]), float(g["decay"]), g["E"], dev)
    m.train(bool(g["training"]))
    q, loss, (perp, idx, enc) = m(torch.from_numpy(z).to(dev), temperature=float(g["T"]))
    # tuple layout of vqvae/ema_vector_quantizer.py:108
    assert q.shape == z.shape and loss.dim() == 0 and perp.dim() == 0
    assert idx.shape == (z.size // D, 1) and idx.dtype == torch.int64
    assert enc.shape == (z.size // D, K) and enc.dtype == torch.float32
    assert np.array_equal(idx.cpu().numpy(), g["indices"])
    assert torch.equal(enc.argmax(1, keepdim=True), idx) and float(enc.sum()) == z.size // D
    assert rel_err(q.detach().cpu().numpy(), g["quantized"]) <= TOL
    assert abs(loss.item() - g["loss"]) <= TOL * abs(g["loss"])
    assert abs(perp.item() - g["perplexity"]) <= TOL * abs(g["perplexity"])
    sd = m.state_dict()
    assert list(sd.keys()) == ["ema_cluster_size", "ema_w", "usage_count", "last_reset_epoch",
                               "last_low_usage_reset_epoch", "embedding.weight"]
    for key in sd:
        ref = g["state_" + key.replace(".", "_")]
        assert rel_err(sd[key].cpu().numpy(), ref) <= TOL, key
    s = m.get_codebook_stats()
    ref = g["stats"]
    assert s["active_size"] == int(ref[0]) and abs(s["utilization"] - ref[1]) < 1e-12
    assert abs(s["entropy"] - ref[2]) <= 1e-5 * max(abs(ref[2]), 1e-3)
    assert abs(s["normalized_entropy"] - ref[3]) <= 1e-5 * max(abs(ref[3]), 1e-3)
    assert abs(s["total_usage"] - ref[4]) <= 1e-6 * max(abs(ref[4]), 1.0)
    assert isinstance(s["active_size"], int) and isinstance(s["entropy"], (int, float))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g2_*.npz"))))
def test_training_trajectory_matches_reference(path, dev):
    g = np.load(path)
    K, D = g["E0"].shape
    m = make(K, D, float(g["beta"]), float(g["decay"]), g["E0"], dev)
    m.train()
    for step, seed in enumerate(g["z_seeds"]):
        z = gauss(int(seed), (4, D, 8, 8))
        q, loss, (perp, idx, _) = m(torch.from_numpy(z).to(dev), temperature=float(g["T"]))
        assert np.array_equal(idx.cpu().numpy(), g[f"s{step}_indices"]), f"step {step}"
        assert abs(loss.item() - g[f"s{step}_loss"]) <= TOL * abs(g[f"s{step}_loss"])
        assert abs(perp.item() - g[f"s{step}_perplexity"]) <= TOL * abs(g[f"s{step}_perplexity"])
        assert abs(q.double().sum().item() - g[f"s{step}_quantized_sum"]) <= TOL * q.abs().double().sum().item()
        assert rel_err(m.ema_cluster_size.cpu().numpy(), g[f"s{step}_state_ema_cluster_size"]) <= TOL
        assert rel_err(m.ema_w.cpu().numpy(), g[f"s{step}_state_ema_w"]) <= TOL
        assert rel_err(m.embedding.weight.detach().cpu().numpy(), g[f"s{step}_state_embedding_weight"]) <= TOL
        assert np.array_equal(m.usage_count.cpu().numpy(), g[f"s{step}_state_usage_count"])


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_backward_matches_reference(mode, dev):
    g = np.load(os.path.join(GOLDEN, f"g4_backward_{mode}.npz"))
    K, D = g["E"].shape
    m = make(K, D, float(g["beta"]), 0.95, g["E"], dev)
    m.train(mode == "train")
    z = torch.from_numpy(g["z"]).to(dev).requires_grad_(True)
    q, loss, _ = m(z)
    obj = float(g["a"]) * (q * torch.from_numpy(g["w"]).to(dev)).sum() + float(g["b"]) * loss
    obj.backward()
    assert abs(loss.item() - g["loss"]) <= TOL * abs(g["loss"])
    assert rel_err(z.grad.cpu().numpy(), g["grad"]) <= TOL
    assert m.embedding.weight.grad is None          # nothing flows to the codebook (reference :92, :95)


def test_eval_under_no_grad_and_outer_no_grad_training(dev):
    """The EMA branch keys on module.training, not on grad mode (reference :61): a training-mode
    forward inside torch.no_grad() still updates the codebook."""
    K, D = 128, 32
    E = gauss(1, (K, D)) * np.float32(0.5)
    m = make(K, D, 3.0, 0.95, E, dev)
    z = torch.from_numpy(gauss(2, (4, D, 8, 8))).to(dev)
    m.train()
    with torch.no_grad():
        m(z)
    assert float(m.usage_count.sum()) == 4 * 64
    before = m.embedding.weight.detach().clone()
    m.eval()
    m(z)
    assert torch.equal(before, m.embedding.weight.detach())


def test_state_dict_round_trip_and_strict_load(dev):
    K, D = 256, 64
    m = make(K, D, 3.0, 0.95, gauss(3, (K, D)), dev)
    m.train()
    z = torch.from_numpy(gauss(4, (4, D, 8, 8))).to(dev)
    m(z)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    from vq_vae_amd import EMAVectorQuantizer
    m2 = EMAVectorQuantizer(K, D, 3.0).to(dev)
    m2.load_state_dict(sd, strict=True)
    m2.train()
    o1 = m(z)
    o2 = m2(z)
    assert torch.equal(o1[2][1], o2[2][1]) and torch.equal(o1[0], o2[0])
    assert torch.equal(m.embedding.weight, m2.embedding.weight)


def test_lazy_onehot_above_threshold(dev):
    from vq_vae_amd import EMAVectorQuantizer, LazyOneHot
    K, D = 512, 64
    m = make(K, D, 3.0, 0.95, gauss(5, (K, D)), dev)
    m.onehot_eager_bytes = 1024           # force the lazy form
    m.eval()
    z = torch.from_numpy(gauss(6, (2, D, 8, 8))).to(dev)
    q, loss, (perp, idx, enc) = m(z)
    assert isinstance(enc, LazyOneHot) and enc.shape == (128, K) and enc._dense is None
    flat = enc.detach().flatten()          # what vqgan_trainer.py:143 does with "indices"
    assert flat.shape == (128 * K,) and float(flat.sum()) == 128
    assert torch.equal(torch.argmax(enc, dim=1), idx.view(-1))     # torch function protocol
    assert enc.flatten().cpu().numpy().shape == (128 * K,)         # codebook_utilization_check.py:79
    # operators a dense tensor supports: indexing, len, iteration, arithmetic, comparison, matmul
    q2, _, (_, idx2, lazy) = m(z)
    assert isinstance(lazy, LazyOneHot) and lazy._dense is None
    assert len(lazy) == 128 and int(torch.argmax(lazy[3])) == int(idx2[3]) and lazy._dense is None
    dense = enc.materialize()
    assert torch.equal(lazy[5:9, :], dense[5:9, :])
    assert torch.equal(lazy + 1, dense + 1) and torch.equal(2.0 * lazy, dense * 2.0)
    assert torch.equal(lazy @ m.embedding.weight, dense @ m.embedding.weight)
    assert bool((lazy == dense).all()) and torch.equal(-lazy, -dense)
    assert sum(1 for _ in lazy) == 128


def test_half_and_noncontiguous_inputs(dev):
    K, D = 128, 32
    E = gauss(7, (K, D)) * np.float32(0.5)
    m = make(K, D, 3.0, 0.95, E, dev)
    m.eval()
    z = torch.from_numpy(gauss(8, (2, D, 8, 8))).to(dev)
    ref = m(z)
    nhwc = z.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # channels-last strides
    assert not nhwc.is_contiguous()
    out = m(nhwc)
    assert torch.equal(out[2][1], ref[2][1]) and torch.equal(out[0], ref[0])
    with torch.autocast("cuda", dtype=torch.float16):                   # validation runs under autocast
        out16 = m(z)
    assert torch.equal(out16[2][1], ref[2][1])                          # the search stays fp32
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, D + 1, 4, 4, device=dev))


def test_maintenance_methods(dev):
    K, D = 256, 32
    m = make(K, D, 3.0, 0.95, gauss(9, (K, D)), dev)
    m.train()
    for s in range(3):
        m(torch.from_numpy(gauss(20 + s, (2, D, 8, 8))).to(dev))
    torch.manual_seed(0)
    cs = m.ema_cluster_size.clone()
    n_dead = m.reset_dead_codes(threshold=2, current_epoch=5)
    assert n_dead == int((cs < 2).sum())
    assert float(m.last_reset_epoch) == 5 and m.reset_dead_codes(threshold=2, current_epoch=6) == 0
    dead = (cs < 2).nonzero().view(-1)
    assert float(m.usage_count[dead].abs().sum()) == 0
    n_low = m.reset_low_usage_codes(percentage=0.1, current_epoch=5)
    assert n_low == max(1, int(K * 0.1)) and float(m.last_low_usage_reset_epoch) == 5
    assert m.expand_codebook(K + 64) and m.embedding.weight.shape == (K + 64, D)
    assert m.embedding.weight.is_cuda and m.ema_w.shape == (K + 64, D)
    q, loss, (perp, idx, enc) = m(torch.from_numpy(gauss(30, (2, D, 8, 8))).to(dev))   # still runs after expansion
    assert enc.shape[1] == K + 64 and int(idx.max()) < K + 64


def test_reseed_kernel_semantics(dev):
    """vq_reseed_codes through the C ABI: every target becomes one of the donors plus N(0, noise^2),
    its EMA statistics the donor's times `scale`, usage zero; donors are read before targets are
    written; other rows untouched; the same seed reproduces the result, another seed does not.
    Random-number policy: the reference's distributions (uniform donor, normal noise), not its
    generator sequence -- checked statistically."""
    from scipy import stats as sps
    from vq_vae_amd import ops
    K, D = 512, 256
    E0 = torch.from_numpy(gauss(50, (K, D))).to(dev)
    w0 = torch.from_numpy(gauss(51, (K, D))).to(dev)
    cs0 = torch.rand(K, device=dev) + 1.0
    targets = torch.arange(100, 400, device=dev)
    donors = torch.tensor([5, 6, 7, 150, 420], device=dev)              # 150 is itself a target
    ws = ops.Workspace(dev)

    def run(seed):
        E, w, cs, us = E0.clone(), w0.clone(), cs0.clone(), torch.ones(K, device=dev)
        ops.reseed_codes(E, w, cs, us, targets, donors, 0.25, 0.1, ws, seed=seed)
        torch.cuda.synchronize()
        return E, w, cs, us

    E, w, cs, us = run(1234)
    keep = torch.ones(K, dtype=torch.bool, device=dev); keep[targets] = False
    assert torch.equal(E[keep], E0[keep]) and torch.equal(w[keep], w0[keep]) and torch.equal(cs[keep], cs0[keep])
    assert float(us[targets].abs().sum()) == 0 and float(us[keep].sum()) == int(keep.sum())
    # identify each target's donor from the EMA statistics (exact multiples of the donor's OLD values)
    ratio = cs[targets][:, None] / (cs0[donors][None, :] * np.float32(0.1))
    pick = (ratio - 1).abs().argmin(1)
    assert torch.equal(cs[targets], cs0[donors][pick] * np.float32(0.1))
    assert torch.equal(w[targets], w0[donors][pick] * np.float32(0.1))
    counts = torch.bincount(pick, minlength=len(donors)).cpu().numpy()
    assert counts.min() > 0 and sps.chisquare(counts).pvalue > 1e-4      # uniform over the donors
    noise = ((E[targets] - E0[donors][pick]) / 0.25).cpu().numpy().ravel()   # 76800 samples
    assert abs(noise.mean()) < 0.02 and abs(noise.std() - 1) < 0.02
    assert sps.kstest(noise, "norm").pvalue > 1e-4
    assert abs(sps.skew(noise)) < 0.05 and abs(sps.kurtosis(noise)) < 0.1
    E2, *_ = run(1234)
    assert torch.equal(E, E2)
    E3, *_ = run(1235)
    assert not torch.equal(E, E3)


def test_last_step_counters_report_nonfinite_rows(dev):
    K, D = 256, 32
    m = make(K, D, 3.0, 0.95, gauss(70, (K, D)), dev)
    m.eval()
    z = gauss(71, (2, D, 8, 8))
    m(torch.from_numpy(z).to(dev))
    assert m.last_step_counters()["nonfinite_rows"] == 0
    z[0, :, 1, 1] = np.nan
    z[1, 2, 3, 3] = np.inf
    m(torch.from_numpy(z).to(dev))
    c = m.last_step_counters()
    assert c["nonfinite_rows"] == 2 and not c["codebook_not_finite"]


def test_channels_last_input_is_consumed_without_copy(dev):
    """A channels_last (NHWC in memory) input gives the same results as the NCHW tensor, the
    quantized output comes back in channels_last, and gradients flow."""
    K, D = 256, 64
    E = gauss(80, (K, D)) * np.float32(0.5)
    z = torch.from_numpy(gauss(81, (4, D, 16, 16))).to(dev)
    zl = z.contiguous(memory_format=torch.channels_last)
    assert not zl.is_contiguous()
    ref = make(K, D, 3.0, 0.95, E, dev); ref.train()
    cl = make(K, D, 3.0, 0.95, E, dev); cl.train()
    a = z.clone().requires_grad_(True)
    b = zl.clone(memory_format=torch.preserve_format).requires_grad_(True)
    q0, l0, (p0, i0, _) = ref(a)
    q1, l1, (p1, i1, _) = cl(b)
    assert torch.equal(i0, i1) and torch.equal(q0, q1) and q1.is_contiguous(memory_format=torch.channels_last)
    assert abs(l0.item() - l1.item()) <= 1e-6 * abs(l0.item()) and torch.equal(p0, p1)
    for k, v in ref.state_dict().items():
        assert rel_err(cl.state_dict()[k].cpu().numpy(), v.cpu().numpy()) <= 1e-6, k
    (q0.sum() * 0.5 + l0).backward()
    (q1.sum() * 0.5 + l1).backward()
    assert rel_err(b.grad.cpu().numpy(), a.grad.cpu().numpy()) <= 1e-6


def test_reference_checkpoint_file_on_gpu(dev):
    """The reference-layout checkpoint fixture (written from the reference quantizer) loaded into the GPU
    module: the eval forward reproduces the outputs the reference produced from that state."""
    import os
    from conftest import GOLDEN
    from vq_vae_amd import EMAVectorQuantizer, checkpoint
    ck = checkpoint.load_reference_checkpoint(os.path.join(GOLDEN, "g10_checkpoint_light.pt"))
    g = np.load(os.path.join(GOLDEN, "g10_after_load.npz"))
    m = EMAVectorQuantizer(ck["num_embeddings"], ck["embedding_dim"], 3.0).to(dev)
    m.load_state_dict(ck["quantizer_state"], strict=True)
    m.eval()
    q, loss, (perp, idx, enc) = m(torch.from_numpy(g["z"]).to(dev))
    assert np.array_equal(idx.cpu().numpy().reshape(-1), g["indices"].reshape(-1))
    assert rel_err(q.cpu().numpy(), g["quantized"]) <= 1e-6
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    assert abs(perp.item() - float(g["perplexity"])) <= 1e-5 * float(g["perplexity"])
    s = m.get_codebook_stats()
    got = np.array([s["active_size"], s["utilization"], s["entropy"], s["normalized_entropy"], s["total_usage"]])
    assert np.allclose(got, g["stats"], rtol=1e-5, atol=0)


@pytest.mark.parametrize("training", [True, False])
def test_empty_batch_behaves_like_the_reference(training, dev):
    """A batch without rows (B = 0).  The reference's forward (run on CPU for this test's expectations)
    returns an empty quantized tensor and index list, NaN loss and perplexity, and in training mode
    applies the EMA update with all-zero statistics: cluster sizes and sums decay, usage is unchanged,
    and a fresh codebook (all cluster sizes 0) turns into 0/0."""
    K, D = 64, 16
    E = gauss(1301, (K, D)) * np.float32(0.5)
    m = make(K, D, 3.0, 0.95, E, dev)
    m.train()
    m(torch.from_numpy(gauss(1302, (2, D, 4, 4))).to(dev))          # one real step: a non-trivial state
    st = vo.OracleState(embedding=m.embedding.weight.detach().cpu().numpy().copy(), ema_w=m.ema_w.cpu().numpy().copy(),
                        ema_cluster_size=m.ema_cluster_size.cpu().numpy().copy(),
                        usage_count=m.usage_count.cpu().numpy().copy())
    m.train(training)
    z = torch.zeros(0, D, 4, 4, device=dev, requires_grad=True)
    q, loss, (perp, idx, enc) = m(z)
    assert q.shape == (0, D, 4, 4) and idx.shape == (0, 1) and enc.shape == (0, K)
    assert torch.isnan(loss) and torch.isnan(perp)
    (q.sum() + loss).backward()
    assert z.grad.shape == z.shape
    before = (st.embedding.copy(), st.ema_w.copy(), st.ema_cluster_size.copy(), st.usage_count.copy())
    if training:
        vo.ema_update(np.zeros(K, np.int64), np.zeros((K, D), np.float32), st.usage_count, st.ema_cluster_size, st.ema_w,
                      st.embedding, 0.95, 1e-5)
        assert not np.array_equal(st.ema_cluster_size, before[2])
    assert rel_err(m.ema_cluster_size.cpu().numpy(), st.ema_cluster_size) <= 1e-6
    assert rel_err(m.ema_w.cpu().numpy(), st.ema_w) <= 1e-6
    assert np.allclose(m.embedding.weight.detach().cpu().numpy(), st.embedding, rtol=1e-5, atol=1e-7, equal_nan=True)
    assert np.array_equal(m.usage_count.cpu().numpy(), before[3])
    if training:
        fresh = make(K, D, 3.0, 0.95, E, dev)
        fresh.train()
        fresh(torch.zeros(0, D, 2, 2, device=dev))
        assert torch.isnan(fresh.embedding.weight).all() and float(fresh.ema_cluster_size.abs().sum()) == 0


def test_trainer_call_pattern_matches_reference(dev):
    """g11: what VQGANTrainer does to the quantizer (vqvae/vqgan_trainer.py:339, :86, :136;
    vqvae/custom_vqgan.py:102) -- reset_dead_codes per epoch, per batch one forward under no_grad and one
    with gradients, both in training mode (two EMA updates per batch), backward, get_codebook_stats --
    on the GPU module against the reference's own outputs."""
    g = np.load(os.path.join(GOLDEN, "g11_trainer_pattern.npz"))
    K, D = g["E0"].shape
    m = make(K, D, float(g["beta"]), float(g["decay"]), g["E0"], dev)
    m.train()
    T = float(g["T"])
    step = 0
    for epoch in (1, 2):
        assert m.reset_dead_codes(threshold=2, current_epoch=epoch) == int(g[f"e{epoch}_reset"])
        for b in range(2):
            z = gauss(1110 + step, (2, D, 4, 4))
            w = torch.from_numpy(gauss(1130 + step, (2, D, 4, 4))).to(dev)
            with torch.no_grad():
                _, _, (_, idx1, _) = m(torch.from_numpy(z).to(dev), temperature=T)
            zt = torch.from_numpy(z).to(dev).requires_grad_(True)
            quant, loss, (perp, idx, enc) = m(zt, temperature=T)
            ((quant * w).sum() + loss).backward()
            st = m.get_codebook_stats()
            assert np.array_equal(idx1.cpu().numpy().reshape(-1), g[f"s{step}_idx_nograd"].reshape(-1))
            assert np.array_equal(idx.cpu().numpy().reshape(-1), g[f"s{step}_idx"].reshape(-1))
            assert abs(loss.item() - float(g[f"s{step}_loss"])) <= 1e-5 * float(g[f"s{step}_loss"])
            assert abs(perp.item() - float(g[f"s{step}_perplexity"])) <= 1e-5 * float(g[f"s{step}_perplexity"])
            assert rel_err(zt.grad.cpu().numpy(), g[f"s{step}_grad"]) <= 1e-5
            got = np.array([st["active_size"], st["utilization"], st["entropy"], st["normalized_entropy"], st["total_usage"]])
            assert np.allclose(got, g[f"s{step}_stats"], rtol=1e-5, atol=0)
            for k, v in m.state_dict().items():
                assert rel_err(v.cpu().numpy(), g[f"s{step}_state_" + k.replace(".", "_")]) <= 1e-5, k
            step += 1
