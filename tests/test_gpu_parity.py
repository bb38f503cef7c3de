"""GPU parity: HIP kernels (through the C ABI) against the CPU oracle and the golden vectors.

Bars: indices, histograms and everything integer are bit-exact; with equal indices the
straight-through output and the winning distances are bit-exact too (same fp32 operation
order); sums (loss, EMA statistics, codebook) within 1e-5 of the tensor's scale.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from inputs import gauss, ref_init_codebook
from oracle import vq_oracle as vo

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vq_vae_amd import ops as _ops
    return _ops


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


CASES = [
    # (K, D, z shape) -- NCHW unless 2-D
    (512, 64, (16, 64, 16, 16)),        # benchmark configuration 1
    (8192, 256, (2, 256, 32, 32)),      # headline codebook shape, 2048 rows
    (64, 256, (1, 256, 8, 8)),          # improved_vq_config: K=64
    (100, 32, (3, 32, 5, 7)),           # K not a multiple of 64, HW=35 unaligned, rows % 256 != 0
    (640, 128, (1, 128, 1, 1)),         # a single row
    (8192, 256, (300, 256)),            # row-major latents
    (200, 16, (1000, 16)),
]


@pytest.mark.parametrize("K,D,shape", CASES)
def test_search_bit_exact(K, D, shape, dev, ops):
    z = gauss(11, shape)
    E = gauss(12, (K, D)) * np.float32(0.8)
    ws = ops.Workspace(dev)
    zt, Et = _t(z, dev), _t(E, dev)
    image, e2, control = ops.prepare_codebook(Et, ws)
    assert np.array_equal(_bits(e2[:K].cpu().numpy()), _bits(vo.codebook_norms(E))), "codebook norms differ"
    assert torch.isinf(e2[K:]).all()
    r = ops.search(zt, Et, ws, want_quantized=True, want_dmin=True, prepared=(image, e2, control))
    torch.cuda.synchronize()
    idx_o, dmin_o = vo.search(z, E)
    idx_g = r["indices"].cpu().numpy()
    assert np.array_equal(idx_g, idx_o), f"{np.count_nonzero(idx_g != idx_o)} of {idx_o.size} indices differ"
    assert np.array_equal(r["idx32"][: idx_o.size].cpu().numpy(), idx_o.astype(np.int32))
    assert np.array_equal(_bits(r["dmin"].cpu().numpy()), _bits(dmin_o)), "winning distances differ"
    q_o, sq_o = vo.gather_st(z, E, idx_o)
    assert np.array_equal(_bits(r["quantized"].cpu().numpy()), _bits(q_o)), "straight-through output differs"
    sq_g = float(r["loss_partial"][: r["n_partial"]].double().sum().item())
    assert abs(sq_g - sq_o) <= TOL * sq_o
    # the exact scalar kernel must agree with the MFMA kernel bit for bit
    rx = ops.search(zt, Et, ws, want_quantized=True, want_dmin=True, exact=True, prepared=(image, e2, control))
    assert torch.equal(rx["indices"], r["indices"])
    assert torch.equal(rx["dmin"].view(torch.int32), r["dmin"].view(torch.int32))
    assert torch.equal(rx["quantized"].view(torch.int32), r["quantized"].view(torch.int32))


@pytest.mark.parametrize("K,D,shape", [(8192, 256, (8, 256, 32, 32)), (8192, 256, (3000, 256)),
                                       (1024, 64, (1, 64, 20, 20)), (8192, 256, (1, 256, 3, 3)),
                                       (8192, 256, (65, 256, 32, 32)),      # 260 row tiles: eight parts
                                       (1024, 64, (98304, 64))])            # 384 row tiles: two parts
def test_small_batch_codebook_split(K, D, shape, dev, ops):
    """Batches below 32 Ki rows, and larger ones whose last round of row tiles would run nearly
    empty, split the codebook sweep over extra workgroups; results must be identical to the
    unsplit sweep and to the oracle, non-finite rows included."""
    z = gauss(91, shape)
    E = gauss(92, (K, D)) * np.float32(0.8)
    E[K - 3] = E[5]                                      # duplicate in the last split: lowest index wins
    if len(shape) == 4:
        z[0, :, 0, 0] = E[5]
        z[0, 3, 1, 1] = np.nan
    else:
        z[0] = E[5]
        z[7, 9] = np.inf
    ws = ops.Workspace(dev)
    zt, Et = _t(z, dev), _t(E, dev)
    from vq_vae_amd import _lib
    assert _lib.load().vq_search_split_bytes(zt.numel() // D, K) > 0         # this shape is split
    a = ops.search(zt, Et, ws, want_quantized=True, want_dmin=True, allow_split=True)
    b = ops.search(zt, Et, ws, want_quantized=True, want_dmin=True, allow_split=False)
    assert torch.equal(a["indices"], b["indices"]) and torch.equal(a["idx32"][: a["N"]], b["idx32"][: b["N"]])
    assert torch.equal(a["dmin"].view(torch.int32), b["dmin"].view(torch.int32))
    assert torch.equal(a["quantized"].view(torch.int32), b["quantized"].view(torch.int32))
    sa = float(a["loss_partial"][: a["n_partial"]].double().sum()); sb = float(b["loss_partial"][: b["n_partial"]].double().sum())
    assert (np.isnan(sa) and np.isnan(sb)) or abs(sa - sb) <= 1e-6 * abs(sb) or (np.isinf(sa) and sa == sb)
    idx_o, dmin_o = vo.search(z, E)
    assert np.array_equal(a["indices"].cpu().numpy(), idx_o)
    assert int(idx_o[0]) == 5


@pytest.mark.parametrize("T", [2.5, 0.7])
def test_search_temperature(T, dev, ops):
    K, D = 512, 64
    z = gauss(21, (4, D, 8, 8))
    E = gauss(22, (K, D)) * np.float32(0.5)
    ws = ops.Workspace(dev)
    r = ops.search(_t(z, dev), _t(E, dev), ws, temperature=T, want_dmin=True)
    idx_o, dmin_o = vo.search(z, E, T=T)
    assert np.array_equal(r["indices"].cpu().numpy(), idx_o)
    assert np.array_equal(_bits(r["dmin"].cpu().numpy()), _bits(dmin_o))


def test_ties_and_duplicates(dev, ops):
    g = np.load(os.path.join(GOLDEN, "g3_ties.npz"))
    # D=8 is not an MFMA shape: widen to 16 channels with zeros (distances unchanged)
    E = np.zeros((16, 16), np.float32); E[:, :8] = g["E"]
    z = np.zeros((1, 16, 4, 4), np.float32); z[:, :8] = g["z"]
    ws = ops.Workspace(dev)
    r = ops.search(_t(z, dev), _t(E, dev), ws)
    assert np.array_equal(r["indices"].cpu().numpy(), g["indices"].ravel())
    # native D=8 goes through the exact kernel
    r8 = ops.search(_t(g["z"], dev), _t(g["E"], dev), ws)
    assert np.array_equal(r8["indices"].cpu().numpy(), g["indices"].ravel())


def test_nonfinite_rows_follow_torch_argmin(dev, ops):
    g = np.load(os.path.join(GOLDEN, "g3_nonfinite.npz"))
    ws = ops.Workspace(dev)
    r8 = ops.search(_t(g["z"], dev), _t(g["E"], dev), ws)
    assert np.array_equal(r8["indices"].cpu().numpy(), g["indices"].ravel())
    # the same rows inside an MFMA-shaped problem: flagged rows take the exact path
    K, D = 512, 64
    z = gauss(31, (2, D, 16, 16))
    E = gauss(32, (K, D))
    z[0, 3, 0, 0] = np.nan
    z[0, 5, 0, 1] = np.inf
    z[1, 7, 2, 2] = -np.inf
    z[1, :, 3, 3] = 3e19            # squared norm overflows
    r = ops.search(_t(z, dev), _t(E, dev), ws, want_dmin=True)
    idx_o, dmin_o = vo.search(z, E)
    assert np.array_equal(r["indices"].cpu().numpy(), idx_o)
    assert np.array_equal(_bits(r["dmin"].cpu().numpy()), _bits(dmin_o))
    q_o, _ = vo.gather_st(z, E, idx_o)
    assert np.array_equal(_bits(r["quantized"].cpu().numpy()), _bits(q_o))
    # a codebook with a non-finite row
    E2 = E.copy(); E2[77, 3] = np.nan
    r = ops.search(_t(z, dev), _t(E2, dev), ws)
    idx_o, _ = vo.search(z, E2)
    assert np.array_equal(r["indices"].cpu().numpy(), idx_o)
    # the NaN column wins every row whose own distances are not all NaN, as in torch.argmin
    assert (idx_o == 77).sum() >= idx_o.size - 4 and set(np.unique(idx_o)) <= {0, 77}


@pytest.mark.parametrize("name", ["g3_T-1.0.npz", "g3_T0.0.npz"])
def test_degenerate_temperature(name, dev, ops):
    g = np.load(os.path.join(GOLDEN, name))
    E = np.zeros((16, 16), np.float32); E[:, :8] = g["E"]
    z = np.zeros((1, 16, 4, 4), np.float32); z[:, :8] = g["z"]
    ws = ops.Workspace(dev)
    r = ops.search(_t(z, dev), _t(E, dev), ws, temperature=float(g["T"]))
    assert np.array_equal(r["indices"].cpu().numpy(), g["indices"].ravel())


@pytest.mark.parametrize("K,D,shape", [(512, 64, (16, 64, 16, 16)), (8192, 256, (4, 256, 32, 32)),
                                       (100, 32, (3, 32, 5, 7)), (300, 24, (2, 24, 9, 9))])
def test_histogram_and_embed_sums(K, D, shape, dev, ops):
    z = gauss(41, shape)
    rng = np.random.Generator(np.random.PCG64(42))
    N = z.size // D
    idx = rng.integers(0, K, N, dtype=np.int64)
    idx[: N // 3] = idx[0]                                # a heavy code
    ws = ops.Workspace(dev)
    counts = ops.histogram(_t(idx, dev), K, ws)
    assert np.array_equal(counts.cpu().numpy().view(np.uint32).astype(np.int64), vo.histogram(idx, K))
    counts32 = ops.histogram(_t(idx.astype(np.int32), dev), K, ws)
    assert torch.equal(counts, counts32)
    slabs, nslab = ops.embed_sums(_t(z, dev), _t(idx.astype(np.int32), dev), K, ws)
    dw = ops.reduce_slabs(slabs).cpu().numpy()
    assert rel_err(dw, vo.embed_sum(z, idx, K)) <= TOL
    # reproducible: a second run gives the same bits
    slabs2, _ = ops.embed_sums(_t(z, dev), _t(idx.astype(np.int32), dev), K, ws)
    assert np.array_equal(_bits(ops.reduce_slabs(slabs2).cpu().numpy()), _bits(dw))


@pytest.mark.parametrize("K,D,N", [(8192, 256, 40000), (512, 64, 4096), (100, 32, 777), (16384, 16, 5000),
                                   (64, 4, 3), (700, 128, 70000)])
def test_sorted_row_major_embed_sums(K, D, N, dev, ops):
    """Row-major latents: the counting-sort kernels against the oracle, the code-range kernel and
    themselves (bit-reproducible although the ranking uses atomics); a heavy code, an all-one-code
    batch, a shard's code_base window and a batch holding NaN / inf."""
    z = gauss(43, (N, D))
    rng = np.random.Generator(np.random.PCG64(44))
    idx = rng.integers(0, K, N, dtype=np.int64)
    idx[: N // 3] = idx[0]
    ws = ops.Workspace(dev)
    zt, it = _t(z, dev), _t(idx.astype(np.int32), dev)
    dw, nslab = ops.embed_sums(zt, it, K, ws)
    assert nslab == 1 and dw.shape == (1, K, D)
    got = dw[0].cpu().numpy().copy()
    ref = vo.embed_sum(z, idx, K)
    assert rel_err(got, ref) <= 1e-6
    assert np.array_equal(got[np.bincount(idx, minlength=K) == 0], np.zeros_like(got[np.bincount(idx, minlength=K) == 0]))
    again, _ = ops.embed_sums(zt, it, K, ws)
    assert np.array_equal(_bits(again[0].cpu().numpy()), _bits(got))
    old, _ = ops.embed_sums(zt, it, K, ops.Workspace(dev), sorted_rows=False)
    assert rel_err(ops.reduce_slabs(old).cpu().numpy(), got) <= 1e-6
    # every row on one code
    one = np.full(N, K - 1, np.int32)
    dw1, _ = ops.embed_sums(zt, _t(one, dev), K, ws)
    assert rel_err(dw1[0, K - 1].cpu().numpy(), z.astype(np.float64).sum(0)) <= 1e-6 and float(dw1[0, : K - 1].abs().sum()) == 0
    # a shard's window of codes: rows outside [code_base, code_base + K_loc) are skipped
    if K >= 64:
        lo, K_loc = K // 4, K // 2
        dws, _ = ops.embed_sums(zt, it, K_loc, ws, code_base=lo)
        assert rel_err(dws[0].cpu().numpy(), ref[lo: lo + K_loc]) <= 1e-6
    # NaN / inf in the batch: float accumulation that propagates them like the dense product
    if N >= 100:
        zb = z.copy(); zb[5, 1] = np.nan; zb[N - 1, D - 1] = np.inf
        dwb, _ = ops.embed_sums(_t(zb, dev), it, K, ws)
        gb = dwb[0].cpu().numpy()
        assert np.isnan(gb[idx[5], 1]) and np.isinf(gb[idx[N - 1], D - 1])
        clean = np.ones((K, D), bool); clean[idx[5], 1] = False; clean[idx[N - 1], D - 1] = False
        assert np.allclose(gb[clean], ref[clean], rtol=0, atol=1e-4 * np.abs(ref).max())


def _gpu_state(E, dev):
    K, D = E.shape
    return {"E": _t(E, dev), "ema_w": torch.zeros(K, D, device=dev), "cs": torch.zeros(K, device=dev),
            "usage": torch.zeros(K, device=dev)}


def _check_step(out, o_ref, st_gpu, st_ref):
    assert np.array_equal(out.indices.cpu().numpy(), o_ref["indices"].ravel())
    assert np.array_equal(_bits(out.quantized.cpu().numpy()), _bits(o_ref["quantized"]))
    assert abs(out.loss.item() - o_ref["loss"]) <= TOL * abs(o_ref["loss"])
    assert abs(out.perplexity.item() - o_ref["perplexity"]) <= TOL * abs(o_ref["perplexity"])
    assert np.array_equal(st_gpu["usage"].cpu().numpy(), st_ref.usage_count)
    assert rel_err(st_gpu["cs"].cpu().numpy(), st_ref.ema_cluster_size) <= TOL
    assert rel_err(st_gpu["ema_w"].cpu().numpy(), st_ref.ema_w) <= TOL
    assert rel_err(st_gpu["E"].cpu().numpy(), st_ref.embedding) <= TOL


@pytest.mark.parametrize("K,D,shape,decay,beta,T", [
    (512, 64, (16, 64, 16, 16), 0.95, 3.0, 1.0),
    (64, 256, (2, 256, 16, 16), 0.7, 10.0, 2.5),
    (8192, 256, (8, 256, 32, 32), 0.95, 3.0, 1.0),
])
def test_training_trajectory(K, D, shape, decay, beta, T, dev, ops):
    """Several training steps.  Each step the oracle restarts from the GPU's own state, so the
    index comparison stays bit-exact while the float state is compared with the tolerance; a
    free-running oracle is compared at the end."""
    E0 = gauss(52, (K, D)) * np.float32(0.7)
    ws = ops.Workspace(dev)
    g = _gpu_state(E0, dev)
    free = vo.OracleState.fresh(E0)
    for step in range(4):
        z = gauss(60 + step, shape)
        synced = vo.OracleState(g["E"].cpu().numpy().copy(), g["cs"].cpu().numpy().copy(),
                                g["ema_w"].cpu().numpy().copy(), g["usage"].cpu().numpy().copy())
        o_ref = vo.forward(z, synced, beta, decay, 1e-5, T, True)
        vo.forward(z, free, beta, decay, 1e-5, T, True)
        out = ops.quantize_step(_t(z, dev), g["E"], g["ema_w"], g["cs"], g["usage"], beta=beta, decay=decay,
                                epsilon=1e-5, temperature=T, training=True, ws=ws)
        torch.cuda.synchronize()
        _check_step(out, o_ref, g, synced)
    assert rel_err(g["E"].cpu().numpy(), free.embedding) <= 1e-4
    assert rel_err(g["cs"].cpu().numpy(), free.ema_cluster_size) <= 1e-4


@pytest.mark.parametrize("K,D,shape", [
    (1, 16, (1, 16, 2, 3)),          # a single code: every row picks it
    (2, 8, (3, 8, 1, 5)),            # D = 8: matrix cores at 16 channels, 8 of them zero
    (3, 1, (2, 1, 4, 4)),            # one channel
    (65, 16, (1, 16, 1, 1)),         # one row, one code beyond a stage
    (33, 32, (700, 32)),             # row-major, ragged everything
    (7, 100, (2, 100, 3, 3)),        # channel count outside the kernels' own set: zero-padded to 128
    (1000, 48, (9, 48, 16, 16)),     # 48 channels as 64, a ragged last stage
    (300, 49, (2, 49, 7, 9)),        # an odd channel count
    (64, 200, (500, 200)),           # row-major rows of 800 bytes, padded to 256 channels
    (130, 255, (3, 255, 4, 4)),      # one channel short of 256
    (4096, 128, (1, 128, 16, 17)),   # HW = 272 (not a multiple of 32), codebook split of a small batch
])
def test_edge_shapes_two_training_steps(K, D, shape, dev, ops):
    """Degenerate and ragged shapes through the whole training step, two steps from a fresh state (the
    second one starts from a codebook whose unused rows are zero vectors, ties at distance ||z||^2)."""
    E0 = gauss(1501, (K, D)) * np.float32(0.7)
    ws = ops.Workspace(dev)
    g = _gpu_state(E0, dev)
    for step in range(2):
        z = gauss(1510 + step, shape)
        synced = vo.OracleState(g["E"].cpu().numpy().copy(), g["cs"].cpu().numpy().copy(),
                                g["ema_w"].cpu().numpy().copy(), g["usage"].cpu().numpy().copy())
        if z.ndim == 2:      # rows for the GPU; the oracle's forward takes them as one image of N pixels
            o_ref = vo.forward(np.ascontiguousarray(z.T)[None, :, :, None], synced, 3.0, 0.95, 1e-5, 1.0, True)
            o_ref["quantized"] = np.ascontiguousarray(o_ref["quantized"][0, :, :, 0].T)
        else:
            o_ref = vo.forward(z, synced, 3.0, 0.95, 1e-5, 1.0, True)
        out = ops.quantize_step(_t(z, dev), g["E"], g["ema_w"], g["cs"], g["usage"], beta=3.0, decay=0.95,
                                epsilon=1e-5, temperature=1.0, training=True, ws=ws)
        torch.cuda.synchronize()
        _check_step(out, o_ref, g, synced)
        counts = out.counts.cpu().numpy().view(np.uint32).astype(np.int64)
        assert np.array_equal(counts, np.asarray(o_ref["counts"]).astype(np.int64))


def test_eval_step_leaves_state_alone(dev, ops):
    K, D = 512, 64
    E0 = gauss(72, (K, D)) * np.float32(0.5)
    z = gauss(71, (4, D, 16, 16))
    ws = ops.Workspace(dev)
    g = _gpu_state(E0, dev)
    out = ops.quantize_step(_t(z, dev), g["E"], g["ema_w"], g["cs"], g["usage"], beta=3.0, decay=0.95, epsilon=1e-5,
                            temperature=1.0, training=False, ws=ws)
    st = vo.OracleState.fresh(E0)
    o_ref = vo.forward(z, st, 3.0, 0.95, 1e-5, 1.0, False)
    _check_step(out, o_ref, g, st)
    assert np.array_equal(g["E"].cpu().numpy(), E0)


def test_golden_config1_on_gpu(dev, ops):
    """benchmark configuration 1 against the reference's own outputs (golden vector g6)."""
    g = np.load(os.path.join(GOLDEN, "g6_config1.npz"))
    K, D = 512, 64
    z = gauss(int(g["z_seed"]), (16, D, 16, 16))
    E = gauss(int(g["E_seed"]), (K, D)) * np.float32(g["E_scale"])
    ws = ops.Workspace(dev)
    st = _gpu_state(E, dev)
    out = ops.quantize_step(_t(z, dev), st["E"], st["ema_w"], st["cs"], st["usage"], beta=3.0, decay=0.95,
                            epsilon=1e-5, temperature=1.0, training=True, ws=ws)
    ref_idx = g["indices"].astype(np.int64).ravel()
    assert np.array_equal(out.indices.cpu().numpy(), ref_idx)
    assert abs(out.loss.item() - g["loss"]) <= TOL * abs(g["loss"])
    assert abs(out.perplexity.item() - g["perplexity"]) <= TOL * abs(g["perplexity"])
    q64 = out.quantized.double().cpu().numpy()
    assert abs(q64.sum() - g["quantized_sum"]) <= TOL * g["quantized_abs_sum"]
    assert rel_err(st["E"][:8].cpu().numpy(), g["state_embedding_weight_rows"]) <= TOL
    assert rel_err(st["ema_w"][:8].cpu().numpy(), g["state_ema_w_rows"]) <= TOL
    assert rel_err(st["cs"].cpu().numpy(), g["state_ema_cluster_size"]) <= TOL
    assert np.array_equal(st["usage"].cpu().numpy(), g["state_usage_count"])


def test_reference_init_collapse(dev, ops):
    """The reference's U(-1/K, 1/K) initialisation against N(0,1) latents: nearly every row picks
    the same few codes; after one step every unused code becomes the zero vector."""
    g = np.load(os.path.join(GOLDEN, "g1_refinit_train_T1.0.npz"))
    z = gauss(int(g["z_seed"]), g["quantized"].shape)
    E = g["E"]
    ws = ops.Workspace(dev)
    st = _gpu_state(E, dev)
    out = ops.quantize_step(_t(z, dev), st["E"], st["ema_w"], st["cs"], st["usage"], beta=float(g["beta"]),
                            decay=float(g["decay"]), epsilon=1e-5, temperature=1.0, training=True, ws=ws)
    assert np.array_equal(out.indices.cpu().numpy(), g["indices"].ravel())
    assert rel_err(st["E"].cpu().numpy(), g["state_embedding_weight"]) <= TOL
    unused = g["state_usage_count"] == 0
    assert (st["E"].cpu().numpy()[unused] == 0).all()


def test_documented_ctypes_binding(dev):
    """The ctypes stub printed in the integration notes binds the library as shipped and returns
    the oracle's indices (guards the document against drifting from the ABI)."""
    import os, re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "def nearest_codes" in b][0]
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = {}
        exec(compile(block, "INTEGRATION.md", "exec"), ns)
        z = gauss(301, (2, 64, 8, 8)); E = gauss(302, (512, 64))
        idx = ns["nearest_codes"](torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev))
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    assert np.array_equal(idx.cpu().numpy(), vo.search(z, E)[0])


@pytest.mark.parametrize("K,D,shape", [(1000, 48, (16, 48, 16, 16)), (7, 100, (3, 100, 5, 5)), (300, 49, (700, 49)),
                                       (96, 17, (2, 17, 8, 8)), (513, 130, (4, 130, 6, 6))])
@pytest.mark.parametrize("T", [1.0, 2.5])
def test_zero_padded_channel_counts_run_on_the_matrix_cores(K, D, shape, T, dev, ops):
    """Any embedding_dim up to 256 (vqvae/ema_vector_quantizer.py:11 takes any): counts other than 16 / 32 / 64 /
    128 / 256 run at the next of those with zero channels.  Same indices, distances and outputs, bit for bit, as the
    exact scalar kernel and the oracle."""
    z = gauss(1701, shape)
    E = gauss(1702, (K, D)) * np.float32(0.6)
    ws = ops.Workspace(dev)
    zt, Et = _t(z, dev), _t(E, dev)
    fast = ops.search(zt, Et, ws, temperature=T, want_quantized=True, want_dmin=True)
    lp_fast = float(fast["loss_partial"][: fast["n_partial"]].double().sum())
    i_f, d_f, q_f = fast["indices"].clone(), fast["dmin"].clone(), fast["quantized"].clone()
    exact = ops.search(zt, Et, ws, temperature=T, want_quantized=True, want_dmin=True, exact=True)
    lp_exact = float(exact["loss_partial"][: exact["n_partial"]].double().sum())
    torch.cuda.synchronize()
    assert torch.equal(i_f, exact["indices"])
    assert torch.equal(d_f.view(torch.int32), exact["dmin"].view(torch.int32))
    assert torch.equal(q_f.view(torch.int32), exact["quantized"].view(torch.int32))
    assert abs(lp_fast - lp_exact) <= 1e-6 * abs(lp_exact)
    o_idx, o_d = vo.search(z, E, T)
    assert np.array_equal(i_f.cpu().numpy(), o_idx)
    assert np.array_equal(_bits(d_f.cpu().numpy()), _bits(o_d))
    # the library really took the matrix-core path: the image of a D-channel codebook is laid out at the next size up
    from vq_vae_amd._lib import load
    lib = load()
    dp = next(p for p in (16, 32, 64, 128, 256) if p >= D)
    assert lib.vq_image_bytes(K, D) == lib.vq_image_bytes(K, dp) > 256


def test_temperature_fuzz_against_the_exact_kernel(dev, ops):
    """Random temperatures in (0, 100] (vqvae/ema_vector_quantizer.py:49-50 divides every distance by T), on
    random codebooks and on codebooks of near-duplicates, where rounding the quotient creates ties that the
    lower index must win: the matrix-core kernels against the exact kernel and the oracle, indices and
    quotients bit for bit."""
    rng = np.random.default_rng(1801)
    for trial in range(24):
        K, D = [(8192, 64), (1000, 32), (512, 256), (4096, 128)][trial % 4]
        T = float(np.float32(10.0 ** rng.uniform(-3, 2)))
        z = gauss(1810 + trial, (4, D, 16, 16))
        if trial % 3 == 0:
            # clusters of codes a few ulps apart: their quotients collide
            centers = gauss(1850 + trial, ((K + 15) // 16, D))
            E = np.repeat(centers, 16, axis=0)[:K] * (1 + rng.integers(-3, 4, (K, 1)).astype(np.float32) * np.float32(2 ** -23))
            E = np.ascontiguousarray(E[:K].astype(np.float32))
        else:
            E = gauss(1850 + trial, (K, D)) * np.float32(10.0 ** rng.uniform(-2, 2))
        ws = ops.Workspace(dev)
        zt, Et = _t(z, dev), _t(E, dev)
        fast = ops.search(zt, Et, ws, temperature=T, want_quantized=False, want_dmin=True)
        i_f, d_f = fast["indices"].clone(), fast["dmin"].clone()
        exact = ops.search(zt, Et, ws, temperature=T, want_quantized=False, want_dmin=True, exact=True)
        torch.cuda.synchronize()
        assert torch.equal(i_f, exact["indices"]), (trial, K, D, T)
        assert torch.equal(d_f.view(torch.int32), exact["dmin"].view(torch.int32)), (trial, K, D, T)
    o_idx, o_d = vo.search(z, E, T)          # and the last trial against the oracle
    assert np.array_equal(i_f.cpu().numpy(), o_idx) and np.array_equal(_bits(d_f.cpu().numpy()), _bits(o_d))


@pytest.mark.parametrize("images,K,D,rows_layout", [(64, 8192, 64, False), (64, 4096, 128, True), (64, 1000, 48, False),
                                                    (128, 512, 256, False), (128, 64, 256, True), (128, 1000, 32, False)])
def test_deferred_temperature_on_large_batches(images, K, D, rows_layout, dev, ops):
    """Batches large enough for the undivided sweeps (whole codebook per workgroup, and the persistent kernel at
    K <= 1024) with T != 1: the kernels compare undivided distances, divide the winner only and hand rows with a
    code close above the minimum to the exact kernel.  Against the exact kernel's literal division
    (vqvae/ema_vector_quantizer.py:49-53), indices and quotients bit for bit, on a random codebook and on one of
    near-duplicate clusters where quotients collide and the lower index must win."""
    rng = np.random.default_rng(1900 + K + D)
    ws = ops.Workspace(dev)
    z = gauss(1901 + K, (images, D, 32, 32))
    if rows_layout:
        z = np.ascontiguousarray(z.transpose(0, 2, 3, 1).reshape(-1, D))
    zt = _t(z, dev)
    sent = []
    for clustered, T in ((False, 2.5), (True, 2.5), (True, 0.3), (True, 77.0)):
        if clustered:
            centers = gauss(1950 + K, ((K + 15) // 16, D))
            E = np.repeat(centers, 16, axis=0)[:K] * (1 + rng.integers(-3, 4, (K, 1)).astype(np.float32) * np.float32(2 ** -23))
            E = np.ascontiguousarray(E.astype(np.float32))
        else:
            E = gauss(1950 + K, (K, D))
        Et = _t(E, dev)
        fast = ops.search(zt, Et, ws, temperature=T, want_quantized=True, want_dmin=True)
        sent.append(int(fast["control"][0].item()) + int(fast["control"][6].item()))   # both lists of the exact kernel
        i_f, d_f, q_f = fast["indices"].clone(), fast["dmin"].clone(), fast["quantized"].clone()
        exact = ops.search(zt, Et, ws, temperature=T, want_quantized=True, want_dmin=True, exact=True)
        torch.cuda.synchronize()
        assert torch.equal(i_f, exact["indices"]), (clustered, T)
        assert torch.equal(d_f.view(torch.int32), exact["dmin"].view(torch.int32)), (clustered, T)
        assert torch.equal(q_f.view(torch.int32), exact["quantized"].view(torch.int32)), (clustered, T)
    # a random codebook sends next to nothing to the exact kernel, the clustered ones a good share of the rows
    assert sent[0] <= max(64, images * 1024 // 1000), sent
    assert min(sent[1:]) > 0, sent
    # a few dozen such rows (the exact kernel then shares each row out over several workgroups): a few pairs of
    # codes some units in the last place apart, the copy at the lower index -- latents whose nearest code is the later
    # one of a pair, by less than the quotients can tell apart, are the ones that need the literal division
    E = gauss(1950 + K, (K, D))
    for pair in range(max(1, K // 1024)):
        lo, hi = 7 + 37 * pair, K // 2 + 11 + 53 * pair
        E[lo] = E[hi] * (1 + rng.integers(-2, 3, (D,)).astype(np.float32) * np.float32(2 ** -23))
    z2 = z
    zt2, Et = _t(z2, dev), _t(E, dev)
    for T in (2.5, 0.7):
        fast = ops.search(zt2, Et, ws, temperature=T, want_quantized=True, want_dmin=True)
        n_close = int(fast["control"][6].item())
        i_f, d_f, q_f, lp_f = fast["indices"].clone(), fast["dmin"].clone(), fast["quantized"].clone(), fast["loss_partial"].clone()
        lp = lp_f.double().sum().item()
        exact = ops.search(zt2, Et, ws, temperature=T, want_quantized=True, want_dmin=True, exact=True)
        torch.cuda.synchronize()
        assert n_close >= 1 and (n_close <= 512 or K < 512), n_close      # (K < 512: one workgroup per row whatever the count)
        assert torch.equal(i_f, exact["indices"]) and torch.equal(d_f.view(torch.int32), exact["dmin"].view(torch.int32))
        assert torch.equal(q_f.view(torch.int32), exact["quantized"].view(torch.int32))
        assert abs(lp - exact["loss_partial"].double().sum().item()) <= 1e-6 * abs(lp)
        again = ops.search(zt2, Et, ws, temperature=T, want_quantized=True, want_dmin=True)
        if K >= 512:
            # each such row has a partial of its own whatever workgroup wrote it (their order follows the list's)
            assert torch.equal(torch.sort(again["loss_partial"]).values, torch.sort(lp_f).values)


def _step_buffers(lib, N, K, D, dev, layout, HW):
    """Caller-owned work buffers of vq_quantize_step, sized by the library's own helpers."""
    u8 = lambda n: torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)
    nslab = lib.vq_scatter_slabs_for(N, K, D, layout, HW)
    return {"image": u8(lib.vq_image_bytes(K, D)), "e2": u8(lib.vq_norms_bytes(K)),
            "control": torch.zeros(16, dtype=torch.int32, device=dev), "idx32": torch.empty(N, dtype=torch.int32, device=dev),
            "nf_rows": torch.empty(N, dtype=torch.int32, device=dev), "loss_partial": u8(lib.vq_loss_partial_bytes(N)),
            "split": u8(lib.vq_search_split_bytes(N, K)), "counts": torch.zeros(2, K, dtype=torch.int32, device=dev),
            "slabs": torch.empty(nslab * K * D, dtype=torch.float32, device=dev), "nslab": nslab}


def test_whole_forward_through_one_c_call(dev, ops):
    """vq_quantize_step, bound by hand as a C caller would (plain pointers in a struct): two training steps and
    an eval step of configuration 1 against the oracle's forward (vqvae/ema_vector_quantizer.py:33-108)."""
    import ctypes
    from vq_vae_amd import _lib
    lib = _lib.load()
    K, D, shape = 512, 64, (16, 64, 16, 16)
    N, HW = shape[0] * shape[2] * shape[3], shape[2] * shape[3]
    E0 = gauss(2201, (K, D))
    st = vo.OracleState.fresh(E0)
    E = torch.from_numpy(E0).to(dev)
    ema_w, cluster, usage = torch.zeros(K, D, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    b = _step_buffers(lib, N, K, D, dev, 1, HW)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = _lib.StepArgs()
    a.layout, a.N, a.D, a.HW, a.K = 1, N, D, HW, K
    a.E, a.ema_w, a.cluster_size, a.usage = E.data_ptr(), ema_w.data_ptr(), cluster.data_ptr(), usage.data_ptr()
    a.decay, a.one_minus_decay, a.eps, a.k_eps = ops.ema_scalars(K, 0.95, 1e-5)
    a.beta, a.temperature = 3.0, 1.0
    a.image, a.e2, a.control, a.idx32, a.nf_rows = (b[k].data_ptr() for k in ("image", "e2", "control", "idx32", "nf_rows"))
    a.loss_partial, a.split_ws, a.slabs, a.nslab, a.sort_work = b["loss_partial"].data_ptr(), b["split"].data_ptr(), b["slabs"].data_ptr(), b["nslab"], None
    a.dmin = None
    for step, training in enumerate((True, True, False)):
        z_np = gauss(2210 + step, shape)
        z = torch.from_numpy(z_np).to(dev)
        idx64 = torch.empty(N, dtype=torch.int64, device=dev)
        q = torch.empty_like(z)
        scal = torch.empty(2, device=dev)
        a.z, a.training = z.data_ptr(), 1 if training else 0
        a.idx64, a.q_out, a.loss_out, a.perplexity_out = idx64.data_ptr(), q.data_ptr(), scal.data_ptr(), scal.data_ptr() + 4
        a.counts, a.counts_next = b["counts"][step % 2].data_ptr(), b["counts"][1 - step % 2].data_ptr()
        a.control_clean = 0 if step == 0 else 1          # the previous vq_quantize_step left both clean
        assert lib.vq_quantize_step(ctypes.byref(a), stream) == 0
        torch.cuda.synchronize()
        r = vo.forward(z_np, st, 3.0, 0.95, 1e-5, 1.0, training)
        assert np.array_equal(idx64.cpu().numpy(), r["indices"].ravel())
        assert np.array_equal(_bits(q.cpu().numpy()), _bits(r["quantized"]))
        assert abs(float(scal[0]) - r["loss"]) <= 1e-5 * abs(r["loss"])
        assert abs(float(scal[1]) - r["perplexity"]) <= 1e-5 * r["perplexity"]
        assert rel_err(E.cpu().numpy(), st.embedding) <= 1e-5 and rel_err(ema_w.cpu().numpy(), st.ema_w) <= 1e-5
        assert rel_err(cluster.cpu().numpy(), st.ema_cluster_size) <= 1e-5
        assert np.array_equal(usage.cpu().numpy(), st.usage_count)


def test_packed_statistics_of_two_ranks_through_the_c_calls(dev, ops):
    """The data-parallel exchange as a C caller drives it: two "ranks" (two batches on the one GPU) search, count and
    scatter, vq_pack_stats packs each rank's statistics, the two buffers are added (what the SUM all-reduce does),
    and each rank finishes with vq_step_finish_reduced + vq_ema_codebook_update_end.  Both replicas end bit-identical
    and equal to the oracle's step on the concatenated batch; each rank's loss is its own batch's."""
    import ctypes
    from vq_vae_amd import _lib
    lib = _lib.load()
    K, D, shape = 512, 64, (8, 64, 16, 16)
    N, HW = shape[0] * shape[2] * shape[3], shape[2] * shape[3]
    E0 = gauss(2301, (K, D))
    zs = [gauss(2310 + r, shape) for r in range(2)]
    st = vo.OracleState.fresh(E0)
    ref = vo.forward(np.concatenate(zs, 0), st, 3.0, 0.95, 1e-5, 1.0, True)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    d, omd, eps, keps = ops.ema_scalars(K, 0.95, 1e-5)
    nfl = lib.vq_packed_stats_floats(K, D, 1)
    ranks = []
    for r in range(2):
        ws = ops.Workspace(dev)
        z = torch.from_numpy(zs[r]).to(dev)
        E = torch.from_numpy(E0).to(dev)
        res = ops.search(z, E, ws, want_quantized=True)
        counts = torch.zeros(K, dtype=torch.int32, device=dev)
        assert lib.vq_code_histogram(p(res["idx32"]), 0, N, K, p(counts), 1, p(res["control"]), stream) == 0
        slabs, nslab = ops.embed_sums(z, res["idx32"], K, ws, zmax_valid=True)
        buf = torch.empty(nfl, dtype=torch.float32, device=dev)
        assert lib.vq_pack_stats(p(slabs), nslab, K, D, p(counts), N, p(buf), stream) == 0
        ranks.append({"ws": ws, "E": E, "ema_w": torch.zeros(K, D, device=dev), "cluster": torch.zeros(K, device=dev), "usage": torch.zeros(K, device=dev),
                      "res": res, "lp": res["loss_partial"].clone(), "n_partial": res["n_partial"], "buf": buf,
                      "idx": res["indices"].clone()})
    reduced = ranks[0]["buf"] + ranks[1]["buf"]              # SUM all-reduce
    losses = []
    for r in ranks:
        buf = reduced.clone()
        scal = torch.empty(2, device=dev)
        counts64 = torch.empty(K, dtype=torch.int64, device=dev)
        ntot = torch.empty(1, dtype=torch.int64, device=dev)
        assert lib.vq_step_finish_reduced(p(r["lp"]), r["n_partial"], N, D, 3.0, ctypes.c_void_p(buf.data_ptr() + K * D * 4), K,
                                          d, omd, eps, keps, p(r["usage"]), p(r["cluster"]), p(scal), ctypes.c_void_p(scal.data_ptr() + 4),
                                          p(counts64), p(ntot), None, None, stream) == 0
        assert lib.vq_ema_codebook_update_end(p(buf), 1, K, D, d, omd, p(r["cluster"]), p(r["ema_w"]), p(r["E"]), None, None, stream) == 0
        torch.cuda.synchronize()
        assert int(ntot) == 2 * N and np.array_equal(counts64.cpu().numpy(), ref["counts"].astype(np.int64))
        assert abs(float(scal[1]) - ref["perplexity"]) <= 1e-5 * ref["perplexity"]
        losses.append(float(scal[0]))
    a, b = ranks
    for k in ("E", "ema_w", "cluster", "usage"):
        assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k
    assert rel_err(a["E"].cpu().numpy(), st.embedding) <= 1e-5 and rel_err(a["ema_w"].cpu().numpy(), st.ema_w) <= 1e-5
    assert rel_err(a["cluster"].cpu().numpy(), st.ema_cluster_size) <= 1e-5
    assert np.array_equal(a["usage"].cpu().numpy(), st.usage_count)
    assert np.array_equal(np.concatenate([a["idx"].cpu().numpy(), b["idx"].cpu().numpy()]), ref["indices"].ravel())
    assert abs(sum(losses) / 2 - ref["loss"]) <= 1e-5 * abs(ref["loss"])      # equal batches: the mean of the two local means
