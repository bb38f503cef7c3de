"""Filtered search (bf16 matrix-core filter + exact fp32 verification): its outputs must equal the
fp32 search's and the oracle's bit for bit -- indices, winning distances, quantized output."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from inputs import gauss
from oracle import vq_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _both(z, E, dev, T=1.0):
    from vq_vae_amd import ops
    ws = ops.Workspace(dev)
    zt, Et = torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev)
    out = {}
    for name, filt in (("fp32", False), ("filtered", True)):
        r = ops.search(zt, Et, ws, temperature=T, want_dmin=True, filtered=filt)
        torch.cuda.synchronize()
        loss = r["loss_partial"][: r["n_partial"]].double().sum().item()
        out[name] = (r["indices"].cpu().numpy(), r["dmin"].cpu().numpy(), r["quantized"].cpu().numpy(), loss,
                     r["control"].cpu().numpy().copy())
    return out


def _same_bits(a, b):
    return np.array_equal(a.view(np.int32), b.view(np.int32))


@pytest.mark.parametrize("K,D,shape,T", [
    (512, 64, (16, 64, 16, 16), 1.0),          # benchmark config 1
    (1000, 64, (4, 64, 16, 16), 1.0),          # K not a multiple of the 128-code stage
    (512, 128, (5000, 128), 1.0),              # (N, D) layout, ragged row count
    (640, 64, (4999, 64), 1.0),                # row count not a multiple of 4
    (8192, 256, (8, 256, 32, 32), 1.0),        # benchmark config 2 shape
    (8192, 256, (8, 256, 32, 32), 2.5),        # temperature
    (300, 256, (1, 256, 7, 9), 0.7),
])
def test_filtered_equals_oracle(dev, K, D, shape, T):
    z = gauss(400 + K + D, shape)
    E = gauss(401 + K, (K, D))
    o = _both(z, E, dev, T)
    idx_o, dmin_o = vo.search(z, E, T=T)
    for name in ("fp32", "filtered"):
        idx, dmin, q, loss, ctl = o[name]
        assert np.array_equal(idx, idx_o), name
        assert _same_bits(dmin, np.asarray(dmin_o, np.float32)), name
    assert _same_bits(o["fp32"][2], o["filtered"][2])
    assert abs(o["fp32"][3] - o["filtered"][3]) <= 1e-6 * abs(o["fp32"][3])


def test_filtered_close_codes_and_ties(dev):
    """Clustered latents, duplicated codes (exact ties -> lowest index) and a codebook whose codes
    are too close for the filter (reference-style U(-1/K, 1/K) init -> fp32 fall-back)."""
    K, D = 2048, 64
    E = gauss(410, (K, D))
    rng = np.random.default_rng(5)
    z = (E[rng.integers(0, K, 8192)] + np.float32(0.05) * gauss(411, (8192, D))).astype(np.float32)
    Edup = E.copy(); Edup[K // 2:] = Edup[: K // 2]
    Eref = ((np.random.default_rng(6).random((K, D), dtype=np.float32) * 2 - 1) / K).astype(np.float32)
    for Ecase, expect_fallback in ((E, False), (Edup, None), (Eref, True)):
        o = _both(z, Ecase, dev)
        idx_o, dmin_o = vo.search(z, Ecase)
        assert np.array_equal(o["filtered"][0], idx_o)
        assert _same_bits(o["filtered"][1], np.asarray(dmin_o, np.float32))
        assert _same_bits(o["filtered"][2], o["fp32"][2])
        if expect_fallback is not None:
            unsettled = int(o["filtered"][4][4])           # rows left after the first pass
            assert (unsettled * 8 > z.shape[0]) == expect_fallback
    assert idx_o.max() < K


def test_filtered_nonfinite_rows(dev):
    """NaN / inf / huge / all-zero rows leave the filtered path for the exact kernel and come out
    exactly as from the fp32 search (torch.argmin's NaN rule included)."""
    K, D = 512, 64
    E = gauss(420, (K, D))
    z = gauss(421, (8, D, 16, 16))
    z[0, :, 0, 0] = np.nan
    z[1, 3, 2, 2] = np.inf
    z[2, :, 4, 4] = 0.0
    z[3, :, 5, 5] = 1e20
    z[4, 7, 1, 1] = -np.inf
    o = _both(z, E, dev)
    assert np.array_equal(o["fp32"][0], o["filtered"][0])
    assert np.array_equal(o["fp32"][2].view(np.int32), o["filtered"][2].view(np.int32))
    idx_o, _ = vo.search(z, E)
    assert np.array_equal(o["filtered"][0], idx_o)
    assert int(o["filtered"][4][0]) >= 5                   # at least those five rows took the exact kernel


def test_filtered_module_step_matches_fp32(dev):
    """A training step of the module in 'filtered' mode leaves exactly the state of 'fp32' mode."""
    from vq_vae_amd import EMAVectorQuantizer
    K, D = 1024, 64
    E = gauss(430, (K, D)) * np.float32(0.5)
    z = torch.from_numpy(gauss(431, (64, D, 32, 32))).to(dev)        # 65536 rows, 2^32 work: lower the work bar
    res = {}
    for mode in ("fp32", "filtered"):
        m = EMAVectorQuantizer(K, D, 3.0).to(dev)
        m.search_mode = mode
        m.filtered_min_work = 1 << 30
        assert m._use_filter(z) == (mode == "filtered")
        with torch.no_grad():
            m.embedding.weight.copy_(torch.from_numpy(E))
        m.train()
        for _ in range(2):
            q, loss, (perp, idx, _) = m(z, temperature=1.0)
        torch.cuda.synchronize()
        res[mode] = (q, loss, perp, idx, {k: v.clone() for k, v in m.state_dict().items()})
    a, b = res["fp32"], res["filtered"]
    assert torch.equal(a[3], b[3]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert abs(a[1].item() - b[1].item()) <= 1e-6 * abs(a[1].item())
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    with pytest.raises(ValueError):
        m.search_mode = "bf16"
        m(z)


def test_filtered_unsupported_shapes_use_fp32(dev):
    """D outside {64,128,256} or K > 8192: ops.search silently runs the fp32 search; the C entry
    point itself reports VQ_ERR_UNSUPPORTED."""
    from vq_vae_amd import ops, _lib
    z = gauss(440, (300, 32)); E = gauss(441, (128, 32))
    ws = ops.Workspace(dev)
    r = ops.search(torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev), ws, filtered=True)
    assert np.array_equal(r["indices"].cpu().numpy(), vo.search(z, E)[0])
    assert not ops.filter_supported(16384, 256, 1.0, 0, True)
    assert not ops.filter_supported(512, 64, 0.0, 0, True)
    lib = _lib.load()
    assert lib.vq_filter_image_bytes(16384, 256) == 256


def test_filtered_shard_search_with_code_base(dev):
    """A codebook-row shard searched with code_base (no quantized output): winners, distances and the
    packed (distance, index) keys of the filtered search equal the fp32 search's."""
    from vq_vae_amd import ops
    K, D = 2048, 64
    E = gauss(450, (K, D)); z = gauss(451, (40, D, 32, 32))
    zt = torch.from_numpy(z).to(dev)
    ws = ops.Workspace(dev)
    for lo in (0, 1024):
        Es = torch.from_numpy(E[lo:lo + 1024].copy()).to(dev)
        a = ops.search(zt, Es, ws, code_base=lo, want_quantized=False, want_dmin=True)
        pa = ops.pack_dist_index(a["dmin"], a["indices"]).clone()
        b = ops.search(zt, Es, ws, code_base=lo, want_quantized=False, want_dmin=True, filtered=True)
        pb = ops.pack_dist_index(b["dmin"], b["indices"])
        assert torch.equal(pa, pb)
        assert int(b["indices"].min()) >= lo and int(b["indices"].max()) < lo + 1024


def test_filtered_large_codebook_in_ranges(dev):
    """K above the 8192 codes of one filtered launch: ranges of rows are searched one after the
    other and merged through the packed keys; the result equals the fp32 search of the whole
    codebook, duplicates across ranges and a NaN row included."""
    from vq_vae_amd import ops
    K, D = 20000, 64
    E = gauss(460, (K, D))
    E[15000] = E[77]                                       # exact tie across two ranges -> lowest index
    z = gauss(461, (16, D, 32, 32))
    z[0, :, 0, 0] = E[77]
    z[1, 4, 2, 2] = np.nan
    zt, Et = torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev)
    ws = ops.Workspace(dev)
    a = ops.search(zt, Et, ws, want_dmin=True)
    ai, ad, aq = a["indices"].clone(), a["dmin"].clone(), a["quantized"].clone()
    b = ops.search(zt, Et, ws, want_dmin=True, filtered=True)
    torch.cuda.synchronize()
    assert torch.equal(ai, b["indices"])
    assert torch.equal(ad.view(torch.int32), b["dmin"].view(torch.int32))
    assert torch.equal(aq.view(torch.int32), b["quantized"].view(torch.int32))
    assert int(ai[0]) == 77
    idx_o, _ = vo.search(z, E)
    assert np.array_equal(b["indices"].cpu().numpy(), idx_o)


def test_filtered_is_translation_invariant(dev):
    """A large common offset of latents and codes (5 sigma per channel) must not push the batch to
    the fp32 fall-back: the filter centres its operands on the codebook mean."""
    from vq_vae_amd import ops
    K, D = 2048, 64
    mu = (5.0 * gauss(470, (D,))).astype(np.float32)
    E = gauss(471, (K, D)) + mu
    z = gauss(472, (64, D, 32, 32)) + mu.reshape(1, D, 1, 1)
    o = _both(z, E, dev)
    idx_o, dmin_o = vo.search(z, E)
    assert np.array_equal(o["filtered"][0], idx_o)
    assert _same_bits(o["filtered"][1], np.asarray(dmin_o, np.float32))
    assert _same_bits(o["filtered"][2], o["fp32"][2])
    unsettled = int(o["filtered"][4][4])
    assert unsettled * 8 <= z.shape[0] * 32 * 32, unsettled      # no fall-back


@pytest.mark.parametrize("T", [1.0, 2.5])
def test_a_handful_of_open_rows_goes_straight_to_the_per_row_kernel(dev, T):
    """With a codebook that has followed the data (the EMA cluster means after a training step) the first level
    leaves a dozen rows of a 64 Ki-row batch open; at most 32 such rows skip the second level and are evaluated
    against every code, shared out over several workgroups per row.  Same indices, distances and outputs as the
    fp32 search, bit for bit."""
    from vq_vae_amd import ops
    K, D = 8192, 256
    g = torch.Generator(device=dev).manual_seed(77)
    z = torch.randn(64, D, 32, 32, device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g)
    w, c, u = torch.zeros(K, D, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    ws = ops.Workspace(dev)
    for step in range(10):      # the codebook moves to the cluster means; the open rows dwindle
        ops.quantize_step(z, E, w, c, u, beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=True, ws=ws)
        b = ops.search(z, E, ws, temperature=T, want_dmin=True, filtered=True)
        ctl = b["control"].cpu().numpy()
        if 1 <= ctl[4] <= 32:
            break
    assert 1 <= ctl[4] <= 32 and ctl[5] == 0, ctl[:8]      # open after the first level; the second level stood aside
    ib, db, qb = b["indices"].clone(), b["dmin"].clone(), b["quantized"].clone()
    lb = b["loss_partial"][: b["n_partial"]].double().sum().item()
    assert int((b["idx32"] < 0).sum()) == 0
    a = ops.search(z, E, ws, temperature=T, want_dmin=True, filtered=False)
    torch.cuda.synchronize()
    la = a["loss_partial"][: a["n_partial"]].double().sum().item()
    assert torch.equal(a["indices"], ib)
    assert torch.equal(a["dmin"].view(torch.int32), db.view(torch.int32))
    assert torch.equal(a["quantized"].view(torch.int32), qb.view(torch.int32))
    assert abs(la - lb) <= 1e-6 * abs(la)
