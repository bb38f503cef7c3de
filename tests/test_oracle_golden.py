"""Pins the CPU oracle (oracle/vq_oracle.c) against golden vectors captured from the
reference quantizer (tests/golden/make_golden.py).  CPU only."""
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from inputs import gauss
from oracle import vq_oracle as vo

TOL = 1e-5  # relative to tensor scale; the bar the path's floating-point outputs must meet


def classify_mismatches(z, E, idx_a, idx_b, T=1.0):
    """Rows where two index vectors differ are acceptable only as near-ties: the two
    candidates' distances, recomputed in float64, differ by no more than a few fp32 ulps
    of the distance scale ||z||^2 + ||e||^2."""
    B, D, H, W = z.shape
    zf = np.transpose(z, (0, 2, 3, 1)).reshape(-1, D).astype(np.float64)
    bad = np.nonzero(idx_a.ravel() != idx_b.ravel())[0]
    near = 0
    for n in bad:
        ea, eb = E[idx_a.ravel()[n]].astype(np.float64), E[idx_b.ravel()[n]].astype(np.float64)
        da, db = np.sum((zf[n] - ea) ** 2), np.sum((zf[n] - eb) ** 2)
        scale = np.sum(zf[n] ** 2) + max(np.sum(ea ** 2), np.sum(eb ** 2))
        if abs(da - db) <= 8 * np.finfo(np.float32).eps * scale / abs(T):
            near += 1
    return len(bad), near


def load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_*.npz"))))
def test_single_step(path):
    g = np.load(path)
    z = gauss(int(g["z_seed"]), g["quantized"].shape)
    E = g["E"]
    st = vo.OracleState.fresh(E)
    out = vo.forward(z, st, float(g["beta"]), float(g["decay"]), 1e-5, float(g["T"]), bool(g["training"]))
    nbad, near = classify_mismatches(z, E, out["indices"], g["indices"], float(g["T"]))
    assert nbad == near, f"{nbad - near} index mismatches that are not near-ties"
    if nbad == 0:
        assert rel_err(out["quantized"], g["quantized"]) <= TOL
        assert abs(out["loss"] - g["loss"]) <= TOL * abs(g["loss"])
        assert abs(out["perplexity"] - g["perplexity"]) <= TOL * abs(g["perplexity"])
        assert rel_err(st.embedding, g["state_embedding_weight"]) <= TOL
        assert rel_err(st.ema_w, g["state_ema_w"]) <= TOL
        assert rel_err(st.ema_cluster_size, g["state_ema_cluster_size"]) <= TOL
        assert np.array_equal(st.usage_count, g["state_usage_count"])
        s = vo.cluster_stats(st.ema_cluster_size, st.usage_count)
        ref = g["stats"]
        assert s["active_size"] == int(ref[0])
        assert abs(s["utilization"] - ref[1]) < 1e-12
        assert abs(s["entropy"] - ref[2]) <= 1e-5 * max(abs(ref[2]), 1e-3)
        assert abs(s["normalized_entropy"] - ref[3]) <= 1e-5 * max(abs(ref[3]), 1e-3)
        assert abs(s["total_usage"] - ref[4]) <= 1e-6 * max(abs(ref[4]), 1.0)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g2_*.npz"))))
def test_trajectory(path):
    """Five training steps: the Laplace-smoothed cluster sizes are stored back each step and
    codes unused in step 1 collapse to zero rows; both must show up in the oracle's state."""
    g = np.load(path)
    st = vo.OracleState.fresh(g["E0"])
    K, D = g["E0"].shape
    for step, seed in enumerate(g["z_seeds"]):
        z = gauss(int(seed), (4, D, 8, 8))
        E_before = st.embedding.copy()
        out = vo.forward(z, st, float(g["beta"]), float(g["decay"]), 1e-5, float(g["T"]), True)
        nbad, near = classify_mismatches(z, E_before, out["indices"], g[f"s{step}_indices"], float(g["T"]))
        assert nbad == 0, f"step {step}: {nbad} index mismatches ({near} near-ties) - trajectory diverged"
        assert abs(out["loss"] - g[f"s{step}_loss"]) <= TOL * abs(g[f"s{step}_loss"])
        assert abs(out["perplexity"] - g[f"s{step}_perplexity"]) <= TOL * abs(g[f"s{step}_perplexity"])
        assert rel_err(st.ema_cluster_size, g[f"s{step}_state_ema_cluster_size"]) <= TOL
        assert rel_err(st.ema_w, g[f"s{step}_state_ema_w"]) <= TOL
        assert rel_err(st.embedding, g[f"s{step}_state_embedding_weight"]) <= TOL
        assert np.array_equal(st.usage_count, g[f"s{step}_state_usage_count"])
    unused = g["s0_state_usage_count"] == 0
    if unused.any():
        assert np.all(g["s0_state_embedding_weight"][unused] == 0.0)
        st0 = vo.OracleState.fresh(g["E0"])
        vo.forward(gauss(int(g["z_seeds"][0]), (4, D, 8, 8)), st0, float(g["beta"]), float(g["decay"]), 1e-5,
                   float(g["T"]), True)
        assert np.all(st0.embedding[unused] == 0.0)


def test_ties_lowest_index():
    g = load("g3_ties.npz")
    idx, _ = vo.search(g["z"], g["E"])
    assert np.array_equal(idx, g["indices"].ravel())
    assert idx[0] == 3 and idx[8] == 3          # duplicates of code 3 at rows 5 and 12 never win


def test_nonfinite_rows_follow_torch_argmin():
    g = load("g3_nonfinite.npz")
    idx, _ = vo.search(g["z"], g["E"])
    assert np.array_equal(idx, g["indices"].ravel())


@pytest.mark.parametrize("name", ["g3_T-1.0.npz", "g3_T0.0.npz"])
def test_degenerate_temperature(name):
    g = load(name)
    idx, _ = vo.search(g["z"], g["E"], T=float(g["T"]))
    assert np.array_equal(idx, g["indices"].ravel())


def test_config1_full():
    g = load("g6_config1.npz")
    K, D = 512, 64
    z = gauss(int(g["z_seed"]), (16, D, 16, 16))
    E = gauss(int(g["E_seed"]), (K, D)) * np.float32(g["E_scale"])
    st = vo.OracleState.fresh(E)
    out = vo.forward(z, st, 3.0, 0.95, 1e-5, 1.0, True)
    ref_idx = g["indices"].astype(np.int64)
    assert hashlib.sha256(ref_idx.tobytes()).hexdigest() == str(g["indices_sha256"])
    nbad, near = classify_mismatches(z, E, out["indices"], ref_idx)
    assert nbad == near
    assert nbad == 0
    assert abs(out["loss"] - g["loss"]) <= TOL * abs(g["loss"])
    assert abs(out["perplexity"] - g["perplexity"]) <= TOL * abs(g["perplexity"])
    q64 = out["quantized"].astype(np.float64)
    assert abs(q64.sum() - g["quantized_sum"]) <= TOL * g["quantized_abs_sum"]
    assert abs(np.abs(q64).sum() - g["quantized_abs_sum"]) <= TOL * g["quantized_abs_sum"]
    assert rel_err(st.embedding[:8], g["state_embedding_weight_rows"]) <= TOL
    assert rel_err(st.ema_w[:8], g["state_ema_w_rows"]) <= TOL
    assert abs(st.embedding.astype(np.float64).sum() - g["state_embedding_weight_sum"]) <= TOL * g["state_embedding_weight_abs_sum"]
    assert rel_err(st.ema_cluster_size, g["state_ema_cluster_size"]) <= TOL
    assert np.array_equal(st.usage_count, g["state_usage_count"])


def test_headline_codebook_shape():
    g = load("g8_k8192.npz")
    K, D = 8192, 256
    z = gauss(int(g["z_seed"]), (2, D, 16, 16))
    E = gauss(int(g["E_seed"]), (K, D))
    st = vo.OracleState.fresh(E)
    out = vo.forward(z, st, 3.0, 0.95, 1e-5, 1.0, True)
    nbad, near = classify_mismatches(z, E, out["indices"], g["indices"].astype(np.int64))
    assert nbad == near and nbad <= 2
    if nbad == 0:
        assert abs(out["loss"] - g["loss"]) <= TOL * abs(g["loss"])
        assert abs(out["perplexity"] - g["perplexity"]) <= TOL * abs(g["perplexity"])
        assert rel_err(st.ema_cluster_size, g["state_ema_cluster_size"]) <= TOL
        assert np.array_equal(st.usage_count, g["state_usage_count"])
        assert abs(st.embedding.astype(np.float64).sum() - g["embedding_sum"]) <= TOL * g["embedding_abs_sum"]


def test_dense_port_matches_golden():
    """The dense one-hot torch formulation (the benchmark's CPU baseline) reproduces the
    reference's outputs on golden vector g6 (benchmark configuration 1)."""
    import torch
    from oracle.dense_port import dense_forward, new_state
    g = load("g6_config1.npz")
    K, D = 512, 64
    z = torch.from_numpy(gauss(int(g["z_seed"]), (16, D, 16, 16)))
    E = torch.from_numpy(gauss(int(g["E_seed"]), (K, D)) * np.float32(g["E_scale"]))
    st = new_state(E)
    q, loss, perp, idx = dense_forward(z, st, 3.0)
    assert np.array_equal(idx.numpy().ravel(), g["indices"].astype(np.int64).ravel())
    assert abs(loss.item() - g["loss"]) <= 1e-6 * abs(g["loss"])
    assert abs(perp.item() - g["perplexity"]) <= 1e-6 * abs(g["perplexity"])
    assert rel_err(st["cs"].numpy(), g["state_ema_cluster_size"]) <= 1e-6
    assert rel_err(st["E"][:8].numpy(), g["state_embedding_weight_rows"]) <= 1e-6


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_boundary_convs_around_the_quantizer(mode):
    """quant_conv -> quantizer -> post_quant_conv with gradients (golden g9: the reference quantizer
    between two torch nn.Conv2d(.., 1) layers, composed as vqvae/custom_vqgan.py does)."""
    g = load(f"g9_boundary_{mode}.npz")
    st = vo.OracleState.fresh(g["E"])
    r = vo.boundary(g["h"], g["Wq"], g["bq"], g["Wp"], g["bp"], st, float(g["beta"]), float(g["decay"]), 1e-5, 1.0,
                    mode == "train", grad_out=float(g["a"]) * g["w"], grad_loss=float(g["b"]))
    assert np.array_equal(r["indices"], g["indices"])
    assert rel_err(r["out"], g["out"]) <= TOL
    assert abs(r["loss"] - g["loss"]) <= TOL * abs(g["loss"])
    assert abs(r["perplexity"] - g["perplexity"]) <= TOL * abs(g["perplexity"])
    for name in ("grad_h", "grad_Wq", "grad_bq", "grad_Wp", "grad_bp"):
        assert rel_err(r[name], g[name]) <= TOL, name
    assert rel_err(st.embedding, g["state_embedding_weight"]) <= TOL
    assert rel_err(st.ema_w, g["state_ema_w"]) <= TOL
    assert rel_err(st.ema_cluster_size, g["state_ema_cluster_size"]) <= TOL


def test_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """oracle/asan_check.c: every oracle entry point over awkward sizes, built with
    -fsanitize=address,undefined (host code only)."""
    import shutil
    import subprocess
    odir = os.path.join(os.path.dirname(GOLDEN), "..", "oracle")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    build = subprocess.run(["make", "-s", "-C", odir, "asan_check"], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-200:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", OMP_NUM_THREADS="4")
    run = subprocess.run([os.path.join(odir, "asan_check")], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and "asan_check ok" in run.stdout, run.stderr[-2000:]


def test_trainer_call_pattern(golden_dir):
    """g11: the trainer's use of the quantizer -- two training-mode forwards per batch (the first
    under no_grad), so two EMA updates per batch, T = 2.5, decay 0.7 -- replayed through the oracle."""
    g = np.load(os.path.join(golden_dir, "g11_trainer_pattern.npz"))
    st = vo.OracleState.fresh(g["E0"])
    beta, decay, T = float(g["beta"]), float(g["decay"]), float(g["T"])
    for step in range(4):
        z = gauss(1110 + step, (2, 16, 4, 4))
        o1 = vo.forward(z, st, beta, decay, 1e-5, T, True)
        assert np.array_equal(o1["indices"].reshape(-1), g[f"s{step}_idx_nograd"].reshape(-1))
        o2 = vo.forward(z, st, beta, decay, 1e-5, T, True)
        assert np.array_equal(o2["indices"].reshape(-1), g[f"s{step}_idx"].reshape(-1))
        assert abs(o2["loss"] - g[f"s{step}_loss"]) <= 1e-5 * abs(g[f"s{step}_loss"])
        assert abs(o2["perplexity"] - g[f"s{step}_perplexity"]) <= 1e-5 * g[f"s{step}_perplexity"]
        assert rel_err(st.embedding, g[f"s{step}_state_embedding_weight"]) <= 1e-5
        assert rel_err(st.ema_cluster_size, g[f"s{step}_state_ema_cluster_size"]) <= 1e-5
        assert np.array_equal(st.usage_count, g[f"s{step}_state_usage_count"])
        cs = vo.cluster_stats(st.ema_cluster_size, st.usage_count)
        ref = g[f"s{step}_stats"]
        assert cs["active_size"] == int(ref[0]) and abs(cs["entropy"] - ref[2]) <= 1e-5 * ref[2] and cs["total_usage"] == ref[4]


G12_CASES = ["uniform", "single", "all_once", "half", "wide", "zipf"]


@pytest.mark.parametrize("case", G12_CASES)
def test_index_statistics_match_the_reference_function(case):
    """Golden g12: the reference's own analyze_codebook_utilization (vqvae/codebook_utilization_check.py:64-120)
    run on known index streams; the oracle's float64 statistics reproduce what it returned."""
    g = np.load(f"{GOLDEN}/g12_utilization_reference.npz")
    K, idx = int(g[f"{case}_K"]), g[f"{case}_indices"]
    counts = vo.histogram(idx, K)
    assert np.array_equal(counts.astype(np.float64), g[f"{case}_index_counts"])
    assert np.array_equal(np.flatnonzero(counts), g[f"{case}_unique"])
    st = vo.index_stats_f64(counts, idx.size)
    util, ent, nent, gini = g[f"{case}_scalars"]
    assert st["unique"] == g[f"{case}_unique"].size and st["utilization"] == util
    assert abs(st["entropy"] - ent) <= 1e-12 * max(1.0, abs(ent))
    assert abs(st["normalized_entropy"] - nent) <= 1e-12
    assert abs(st["gini"] - gini) <= 1e-12 * max(1.0, abs(gini))
    assert np.allclose(counts / idx.size, g[f"{case}_probs"], rtol=0, atol=1e-15)


G13_CASES = ["plain", "expand", "reset", "capped", "collapsed"]


@pytest.mark.parametrize("case", G13_CASES)
def test_monitor_codebook_values_match_the_reference_method(case):
    """Golden g13: the reference's own VQGANTrainer.monitor_codebook (vqvae/vqgan_trainer.py:577-650) on the
    reference quantizer after three training steps; the oracle's cumulative statistics (fp32, from the
    stored cluster sizes) and batch statistics (fp32, from the stored last-batch indices) reproduce the
    values the method computed, and its decisions follow from them as restated here."""
    g = np.load(f"{GOLDEN}/g13_monitor_codebook_reference.npz")
    K = g[f"{case}_state_embedding_weight"].shape[0]
    cum = vo.cluster_stats(g[f"{case}_state_ema_cluster_size"], g[f"{case}_state_usage_count"])
    want = g[f"{case}_cumulative"]
    assert cum["active_size"] == int(want[0]) and cum["utilization"] == want[1] and cum["total_usage"] == want[4]
    assert abs(cum["entropy"] - want[2]) <= 2e-6 * want[2] and abs(cum["normalized_entropy"] - want[3]) <= 2e-6
    assert abs(float(g[f"{case}_returned"]) - cum["normalized_entropy"]) <= 2e-6
    idx = g[f"{case}_last_batch_indices"]
    b = vo.index_stats_f32(vo.histogram(idx, K))
    wb = g[f"{case}_batch"]
    assert b["unique"] == int(wb[0]) and abs(b["utilization"] - wb[1]) <= 1e-7
    assert abs(b["entropy"] - wb[2]) <= 2e-6 * wb[2] and abs(b["normalized_entropy"] - wb[3]) <= 2e-6
    epoch, disabled, thr, max_size, pct = g[f"{case}_args"]
    size = K
    if not disabled and cum["utilization"] > thr and K < max_size:          # :622-634
        size = max(K, min(int(K * 1.25), int(max_size)))
    assert size == int(g[f"{case}_size_after"]) and set(g[f"{case}_state_shapes_after"]) == {size}
    last = int(g[f"{case}_state_last_low_usage_reset_epoch"].reshape(-1)[0])
    resets = max(1, int(size * pct)) if epoch > 0 and epoch - last >= 5 else 0  # :641, ema_vector_quantizer.py:245-252
    assert resets == int(g[f"{case}_reset_count"])
    assert int(g[f"{case}_last_reset_epoch_after"]) == (int(epoch) if resets else last)


@pytest.mark.parametrize("case", ["small", "wide", "collapsed", "half"])
def test_used_code_count_matches_the_reference_method(case):
    """Golden g14: the reference's own VQModelTrainer.get_perplexity (vqvae/trainers/vqtrainer.py:46-82) --
    the number of distinct nearest codes of a batch of latents -- against the oracle's search."""
    g = np.load(f"{GOLDEN}/g14_get_perplexity_reference.npz")
    z, E = g[f"{case}_latents"], g[f"{case}_E"]
    rows = np.ascontiguousarray(z.transpose(0, 2, 3, 1).reshape(-1, E.shape[1]))
    idx, _ = vo.search(rows, E)
    assert len(np.unique(idx)) == int(g[f"{case}_used"])


G15_CASES = ["train", "eval", "train_T", "train_2D"]


@pytest.mark.parametrize("case", G15_CASES)
def test_vqgan_forward_methods(case):
    """Golden g15: the reference's own CustomVQGAN.encode / .decode / .forward (vqvae/custom_vqgan.py:62-114)
    around the reference quantizer (identity encoder / decoder): the dict outputs -- with the one-hot
    matrix under "indices" -- the post-step state and all gradients, against the oracle's composition."""
    g = load(f"g15_vqgan_forward_{case}.npz")
    K, D = g["E"].shape
    x = g["x"]
    h = x[:, :D] if x.shape[1] == 2 * D else x                     # the mean half of a 2 D encoder output (:66-67)
    st = vo.OracleState.fresh(g["E"])
    r = vo.boundary(h, g["Wq"], g["bq"], g["Wp"], g["bp"], st, float(g["beta"]), float(g["decay"]), 1e-5, float(g["T"]),
                    case != "eval", grad_out=float(g["a"]) * g["w"], grad_loss=float(g["b"]))
    onehot = g["indices_onehot"]
    assert onehot.shape == (r["indices"].size, K) and np.array_equal(onehot.sum(1), np.ones(onehot.shape[0]))
    assert np.array_equal(r["indices"].reshape(-1), onehot.argmax(1))
    assert rel_err(r["out"], g["decoded_imgs"]) <= TOL
    assert abs(r["loss"] - g["commitment_loss"]) <= TOL * abs(g["commitment_loss"])
    assert abs(r["perplexity"] - g["perplexity"]) <= TOL * abs(g["perplexity"])
    cs = vo.cluster_stats(st.ema_cluster_size, st.usage_count)
    assert abs(cs["entropy"] - g["entropy"]) <= TOL * abs(g["entropy"])
    assert abs(cs["normalized_entropy"] - g["normalized_entropy"]) <= TOL
    gx = g["grad_x"]
    assert rel_err(r["grad_h"], gx[:, :D]) <= TOL and not gx[:, D:].any()
    for name in ("grad_Wq", "grad_bq", "grad_Wp", "grad_bp"):
        assert rel_err(r[name], g[name]) <= TOL, name
    assert rel_err(st.embedding, g["state_embedding_weight"]) <= TOL
    assert rel_err(st.ema_w, g["state_ema_w"]) <= TOL
    assert rel_err(st.ema_cluster_size, g["state_ema_cluster_size"]) <= TOL
    if case == "eval":
        assert np.array_equal(g["tuple_decoded"], g["decoded_imgs"]) and g["tuple_loss"] == g["commitment_loss"]
