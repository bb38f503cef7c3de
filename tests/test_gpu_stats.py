"""GPU tests of the statistics kernels, the packed-key kernels and the row-sharded step."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from inputs import gauss
from oracle import vq_oracle as vo
from test_distributed_cpu import pack_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def numpy_index_stats(a, K):
    """The literal NumPy expressions of vqvae/codebook_utilization_check.py:86-108."""
    uniq = np.unique(a)
    counts = np.bincount(a, minlength=K).astype(np.float64)
    probs = counts / len(a)
    valid = probs[probs > 0]
    entropy = -np.sum(valid * np.log2(valid + 1e-10))
    gini = 1 - np.sum((valid / np.mean(valid)) ** 2) / len(valid)
    return {"utilization": len(uniq) / K, "entropy": entropy, "normalized_entropy": entropy / np.log2(K),
            "gini": gini, "index_counts": counts, "probs": probs, "unique_indices": uniq}


def torch_batch_stats(a, K):
    """The torch fp32 expressions of vqvae/vqgan_trainer.py:594-612 on CPU."""
    t = torch.from_numpy(a)
    counts = torch.bincount(t, minlength=K).float()
    probs = counts / counts.sum()
    valid = probs[probs > 0]
    ent = -torch.sum(valid * torch.log2(valid + 1e-10)).item()
    return {"unique": len(torch.unique(t)), "entropy": ent,
            "normalized_entropy": ent / torch.log2(torch.tensor(float(K))).item()}


def index_sets(K, M, rng):
    zipf = np.minimum(rng.zipf(1.1, M) - 1, K - 1).astype(np.int64)
    return {"uniform": rng.integers(0, K, M, dtype=np.int64), "zipf": zipf,
            "single": np.full(M, K // 3, np.int64), "all_codes": np.arange(M, dtype=np.int64) % K}


@pytest.mark.parametrize("K,M", [(512, 20000), (8192, 1_000_000), (65536, 300_000)])
def test_index_statistics(K, M, dev):
    from vq_vae_amd import stats
    rng = np.random.Generator(np.random.PCG64(7))
    for name, a in index_sets(K, M, rng).items():
        ref = numpy_index_stats(a, K)
        got = stats.analyze_indices(torch.from_numpy(a).to(dev), K)
        assert np.array_equal(got["index_counts"], ref["index_counts"]), name       # integers: exact
        assert np.array_equal(got["unique_indices"], ref["unique_indices"])
        assert got["utilization"] == ref["utilization"]
        for key in ("entropy", "normalized_entropy", "gini"):
            assert abs(got[key] - ref[key]) <= 1e-11 * max(1.0, abs(ref[key])), (name, key)
        o = vo.index_stats_f64(ref["index_counts"].astype(np.int64), M)
        assert abs(o["entropy"] - ref["entropy"]) <= 1e-11 * max(1.0, abs(ref["entropy"]))
        assert abs(o["gini"] - ref["gini"]) <= 1e-11 * max(1.0, abs(ref["gini"]))
        b = stats.batch_code_stats(torch.from_numpy(a).to(dev), K)
        tb = torch_batch_stats(a, K)
        assert b["unique"] == tb["unique"]
        assert abs(b["entropy"] - tb["entropy"]) <= 2e-5 * max(1.0, abs(tb["entropy"]))
        assert abs(b["normalized_entropy"] - tb["normalized_entropy"]) <= 2e-5


def test_accumulating_over_batches(dev):
    from vq_vae_amd import stats
    K = 8192
    rng = np.random.Generator(np.random.PCG64(8))
    parts = [rng.integers(0, K, n, dtype=np.int64) for n in (100_000, 1, 33_333)]
    acc = stats.CodeUsage(K, dev)
    for p in parts:
        acc.update(torch.from_numpy(p).to(dev))
    ref = numpy_index_stats(np.concatenate(parts), K)
    got = acc.result()
    assert np.array_equal(got["index_counts"], ref["index_counts"])
    assert abs(got["entropy"] - ref["entropy"]) <= 1e-11 * ref["entropy"]


def test_count_used_codes(dev):
    from vq_vae_amd import stats
    K, D = 512, 64
    E = gauss(1, (K, D)) * np.float32(0.5)
    z = gauss(2, (8, D, 16, 16))
    idx, _ = vo.search(z, E)
    n = stats.count_used_codes(torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev))
    assert n == len(np.unique(idx))
    flat = np.transpose(z, (0, 2, 3, 1)).reshape(-1, D)                # vqtrainer.py:60 feeds (N, D)
    assert stats.count_used_codes(torch.from_numpy(flat.copy()).to(dev), torch.from_numpy(E).to(dev)) == n


@pytest.mark.parametrize("K,D", [(512, 64), (1000, 32), (8192, 256)])
def test_pairwise_codebook_stats(K, D, dev):
    from vq_vae_amd import stats
    E = gauss(3, (K, D)) * np.float32(0.6)
    E[K // 2] = E[5]                                   # a duplicated code: nearest-other distance 0
    nn_d, nn_i, ref = vo.pairwise(E)
    got = stats.pairwise_codebook_stats(torch.from_numpy(E).to(dev))
    assert np.array_equal(got["nn_idx"].cpu().numpy(), nn_i)
    assert np.array_equal(got["nn_dist"].cpu().numpy().view(np.uint32), nn_d.view(np.uint32))
    assert got["min"] == ref["min"] and got["max"] == ref["max"]
    assert abs(got["mean"] - ref["mean"]) <= 1e-9 * abs(ref["mean"])
    # independent float64 check of the definition ||e_i - e_j||^2
    if K <= 1000:
        e64 = E.astype(np.float64)
        P = ((e64[:, None, :] - e64[None, :, :]) ** 2).sum(-1)
        np.fill_diagonal(P, np.inf)
        assert rel_err(got["nn_dist"].cpu().numpy(), P.min(1)) <= 1e-5
        off = P[np.isfinite(P)]
        assert abs(got["mean"] - off.mean()) <= 1e-5 * off.mean()


def test_pack_unpack_keys(dev):
    from vq_vae_amd import ops
    d = np.array([np.nan, -np.inf, -3.0, -0.0, 0.0, 1e-30, 2.5, np.inf, 7.25], np.float32)
    idx = np.array([5, 4, 3, 2, 1, 0, 8, 7, 6], np.int64)
    packed = ops.pack_dist_index(torch.from_numpy(d).to(dev), torch.from_numpy(idx).to(dev))
    assert np.array_equal(packed.cpu().numpy(), pack_key(d, idx))
    ws = ops.Workspace(dev)
    d2, i64, i32 = ops.unpack_dist_index(packed, ws)
    assert np.array_equal(i64.cpu().numpy(), idx) and np.array_equal(i32[:9].cpu().numpy(), idx.astype(np.int32))
    back = d2.cpu().numpy()
    assert np.isnan(back[0]) and np.array_equal(back[1:], np.where(d[1:] == 0, 0.0, d[1:]).astype(np.float32))


def test_row_sharded_step_equals_unsharded(dev):
    """K = 2048 sharded into 8 x 256 rows on one GPU: the merge is an elementwise int64 min over the
    shards' packed keys (what the MIN all-reduce computes); indices, outputs and the updated
    shards must equal the unsharded step."""
    from vq_vae_amd import ops
    from vq_vae_amd.distributed import shard_rows, sharded_quantize_step
    K, D, P = 2048, 64, 8
    E0 = gauss(4, (K, D)) * np.float32(0.7)
    E0[1500] = E0[100]                                     # duplicate across shards
    z = gauss(5, (8, D, 16, 16))
    zt = torch.from_numpy(z).to(dev)
    # unsharded
    ws = ops.Workspace(dev)
    full = {"E": torch.from_numpy(E0).to(dev), "ema_w": torch.zeros(K, D, device=dev),
            "cs": torch.zeros(K, device=dev), "usage": torch.zeros(K, device=dev)}
    ref = ops.quantize_step(zt, full["E"], full["ema_w"], full["cs"], full["usage"], beta=3.0, decay=0.95,
                            epsilon=1e-5, temperature=1.0, training=True, ws=ws)
    # sharded: phase 1 (search + pack) on every shard, then the merge, then phase 2 per shard
    shards = []
    for r in range(P):
        lo, hi = shard_rows(K, r, P)
        shards.append({"lo": lo, "E": torch.from_numpy(E0[lo:hi].copy()).to(dev),
                       "ema_w": torch.zeros(hi - lo, D, device=dev), "cs": torch.zeros(hi - lo, device=dev),
                       "usage": torch.zeros(hi - lo, device=dev), "ws": ops.Workspace(dev)})
    packs = []
    for s in shards:
        r_ = ops.search(zt, s["E"], s["ws"], code_base=s["lo"], want_quantized=False, want_dmin=True)
        packs.append(ops.pack_dist_index(r_["dmin"], r_["indices"]))
    merged = torch.stack(packs).min(0).values
    E_full = torch.cat([s["E"] for s in shards], 0).clone()
    # Laplace normaliser: sum over shards of the local EMA'd cluster sizes (first step: cs = 0.05 * counts)
    counts = vo.histogram(ref.indices.cpu().numpy(), K)
    n_total = torch.tensor(float(np.float32(1 - 0.95)) * counts.astype(np.float64).sum(), dtype=torch.float64, device=dev)
    outs = []
    for s in shards:
        outs.append(sharded_quantize_step(zt, s["E"], s["ema_w"], s["cs"], s["usage"], K_total=K, code_base=s["lo"],
                                          beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=True,
                                          ws=s["ws"], merge=lambda p: merged.clone(),
                                          gather_codebook=lambda e: E_full, sum_scalar=lambda n: n_total))
    for o in outs:
        assert torch.equal(o.indices, ref.indices)
        assert torch.equal(o.quantized, ref.quantized)
        assert abs(o.loss.item() - ref.loss.item()) <= 1e-6 * abs(ref.loss.item())
        assert abs(o.perplexity.item() - ref.perplexity.item()) <= 1e-6 * ref.perplexity.item()
    assert rel_err(torch.cat([s["E"] for s in shards]).cpu().numpy(), full["E"].cpu().numpy()) <= 1e-5
    assert rel_err(torch.cat([s["cs"] for s in shards]).cpu().numpy(), full["cs"].cpu().numpy()) <= 1e-5
    assert torch.equal(torch.cat([s["usage"] for s in shards]), full["usage"])
    assert torch.equal(torch.cat([s["ema_w"] for s in shards]), full["ema_w"])


def test_data_parallel_hook_over_rccl_single_rank(dev):
    """The data-parallel path end to end on the GPU with a one-rank RCCL group: the all-reduces
    run on device tensors and the step must reproduce the plain step bit for bit."""
    import os
    import torch.distributed as dist
    from vq_vae_amd import ops
    from vq_vae_amd.distributed import make_stats_reduce
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        K, D = 512, 64
        E0 = gauss(81, (K, D)) * np.float32(0.5)
        z = torch.from_numpy(gauss(82, (8, D, 16, 16))).to(dev)
        outs = []
        for hook in (None, make_stats_reduce(), make_stats_reduce(uneven_rows=True)):
            st = {"E": torch.from_numpy(E0).to(dev), "ema_w": torch.zeros(K, D, device=dev),
                  "cs": torch.zeros(K, device=dev), "usage": torch.zeros(K, device=dev)}
            o = ops.quantize_step(z, st["E"], st["ema_w"], st["cs"], st["usage"], beta=3.0, decay=0.95, epsilon=1e-5,
                                  temperature=1.0, training=True, ws=ops.Workspace(dev), stats_reduce=hook)
            outs.append((o, st))
        ref, rst = outs[0]
        for o, st in outs[1:]:
            assert torch.equal(o.indices, ref.indices) and torch.equal(o.quantized, ref.quantized)
            assert o.loss.item() == ref.loss.item() and o.perplexity.item() == ref.perplexity.item()
            for k in rst:
                assert torch.equal(st[k], rst[k]), k
    finally:
        if created:
            dist.destroy_process_group()


def test_analyzer_front_end(dev, tmp_path):
    """analyze_codebook_utilization over a stream of latent batches equals the statistics of the
    oracle's indices for the concatenated batches; the module's state and mode are untouched."""
    import json
    from vq_vae_amd import EMAVectorQuantizer, stats
    K, D = 512, 64
    E = gauss(95, (K, D)) * np.float32(0.5)
    m = EMAVectorQuantizer(K, D, 3.0).to(dev)
    with torch.no_grad():
        m.embedding.weight.copy_(torch.from_numpy(E))
    m.train()
    batches = [gauss(96 + i, (4, D, 16, 16)) for i in range(3)]
    res = stats.analyze_codebook_utilization(m, (torch.from_numpy(b).to(dev) for b in batches), temperature=1.0)
    assert m.training and float(m.usage_count.sum()) == 0          # eval-mode passes, no EMA update
    idx = np.concatenate([vo.search(b, E)[0] for b in batches])
    ref = numpy_index_stats(idx, K)
    assert np.array_equal(res["index_counts"], ref["index_counts"])
    assert res["utilization"] == ref["utilization"] and abs(res["entropy"] - ref["entropy"]) <= 1e-11 * ref["entropy"]
    path = tmp_path / "codebook_stats.json"
    stats.save_codebook_stats(res, str(path))
    saved = json.loads(path.read_text())
    assert saved["n_embed"] == K and len(saved["index_counts"]) == K and len(saved["probs"]) == K


def test_interpolation_smoothness(dev):
    """Quantized interpolation paths: indices equal the oracle's on every interpolated point, the
    path metrics equal a NumPy float64 evaluation of their definitions."""
    from vq_vae_amd import latent_stats
    K, D, P, steps = 512, 64, 40, 12
    E = gauss(1401, (K, D)) * np.float32(0.5)
    a, b = gauss(1402, (P, D)), gauss(1403, (P, D))
    b[3] = a[3]                                                   # a pair with identical ends
    r = latent_stats.interpolation_smoothness(torch.from_numpy(E).to(dev), torch.from_numpy(a).to(dev),
                                              torch.from_numpy(b).to(dev), steps=steps)
    t = torch.linspace(0.0, 1.0, steps + 1, dtype=torch.float32).view(1, steps + 1, 1)
    z = (torch.from_numpy(a).unsqueeze(1) * (1.0 - t) + torch.from_numpy(b).unsqueeze(1) * t).reshape(-1, D).numpy()
    idx, _ = vo.search(z, E)
    idx = idx.reshape(P, steps + 1)
    got = r["indices"].cpu().numpy()
    # the interpolated points are computed on the GPU in fp32: equal up to fp32 rounding of the lerp,
    # which can flip an index only at an exact tie -- none in this data
    assert np.array_equal(got, idx)
    q = E[idx].astype(np.float64)
    step_len = np.linalg.norm(q[:, 1:] - q[:, :-1], axis=2)
    direct = np.linalg.norm(q[:, -1] - q[:, 0], axis=1)
    switches = (idx[:, 1:] != idx[:, :-1]).sum(1)
    assert np.array_equal(r["per_pair"]["code_switches"].cpu().numpy(), switches)
    assert np.array_equal(r["per_pair"]["distinct_codes"].cpu().numpy(), np.array([len(set(row)) for row in idx]))
    ok = direct > 0
    assert abs(r["path_length_ratio"] - (step_len.sum(1)[ok] / direct[ok]).mean()) <= 1e-9
    assert np.isnan(r["per_pair"]["path_length_ratio"].cpu().numpy()[~ok]).all()
    assert switches[3] == 0 and np.isnan(r["per_pair"]["max_jump_ratio"].cpu().numpy()[3])
    assert abs(r["code_switches"] - switches.mean()) <= 1e-12
    qerr = ((z.reshape(P, steps + 1, D).astype(np.float64) - q) ** 2).sum(2).mean(1)
    assert abs(r["quantization_error"] - qerr.mean()) <= 1e-6 * qerr.mean()


def test_monitor_codebook_follows_the_trainer(dev):
    """stats.monitor_codebook against the decisions of VQGANTrainer.monitor_codebook
    (vqvae/vqgan_trainer.py:577-650): batch statistics equal the literal torch expressions, the
    codebook grows by 25 % (capped) once the cumulative utilisation exceeds the threshold, the
    low-usage reset runs for epochs > 0 under the module's own epoch spacing."""
    from vq_vae_amd import EMAVectorQuantizer, stats
    K, D = 64, 16
    m = EMAVectorQuantizer(K, D, 3.0).to(dev)
    with torch.no_grad():
        m.embedding.weight.copy_(torch.from_numpy(gauss(1601, (K, D)) * np.float32(0.5)))
    m.train()
    z = torch.from_numpy(gauss(1602, (4, D, 8, 8))).to(dev)
    _, _, (_, idx, _) = m(z)
    rep = {}
    calls = []
    ne = stats.monitor_codebook(m, idx, 0, max_codebook_size=100, report=rep, on_expand=calls.append, verbose=False)
    tb = torch_batch_stats(idx.cpu().numpy().reshape(-1), K)
    assert rep["batch"]["unique"] == tb["unique"] and abs(rep["batch"]["entropy"] - tb["entropy"]) <= 2e-5 * tb["entropy"]
    # Laplace smoothing makes every cluster size positive after one step: utilisation 1.0 > 0.8 -> expansion
    assert rep["cumulative"]["utilization"] == 1.0 and rep["expanded_to"] == 80 and m.num_embeddings == 80
    assert calls == [m] and m.embedding.weight.shape == (80, D) and m.ema_w.shape == (80, D)
    assert rep["reset_count"] == 0                                   # epoch 0: no reset (:641)
    assert abs(ne - rep["cumulative"]["normalized_entropy"]) == 0
    # the module still runs after the expansion, and the next call is capped at the maximum
    _, _, (_, idx, _) = m(z)
    stats.monitor_codebook(m, idx, 1, max_codebook_size=100, report=rep, verbose=False)
    assert m.num_embeddings == 100 and rep["expanded_to"] == 100
    stats.monitor_codebook(m, idx, 2, max_codebook_size=100, report=rep, verbose=False)
    assert m.num_embeddings == 100 and rep["expanded_to"] is None
    # resets are spaced five epochs apart by the module (vqvae/ema_vector_quantizer.py:245-246)
    stats.monitor_codebook(m, idx, 6, max_codebook_size=100, report=rep, verbose=False)
    assert rep["reset_count"] == 10
    stats.monitor_codebook(m, None, 7, disable_codebook_expansion=True, report=rep, verbose=False)
    assert rep["batch"] is None and rep["reset_count"] == 0


@pytest.mark.parametrize("case", ["uniform", "single", "all_once", "half", "wide", "zipf"])
def test_index_statistics_against_the_reference_function(case, dev):
    """Golden g12 (the reference's analyze_codebook_utilization run on known index streams): the GPU
    statistics, fed in three batches like the reference's loader."""
    from vq_vae_amd import stats
    g = np.load(f"{GOLDEN}/g12_utilization_reference.npz")
    K, idx = int(g[f"{case}_K"]), g[f"{case}_indices"]
    acc = stats.CodeUsage(K, dev)
    for part in np.array_split(idx, 3):
        acc.update(torch.from_numpy(part).to(dev))
    r = acc.result()
    util, ent, nent, gini = g[f"{case}_scalars"]
    assert r["n_embed"] == K and np.array_equal(np.asarray(r["unique_indices"]), g[f"{case}_unique"])
    assert np.array_equal(np.asarray(r["index_counts"], np.float64), g[f"{case}_index_counts"])
    assert r["utilization"] == util
    assert abs(r["entropy"] - ent) <= 1e-12 * max(1.0, abs(ent)) and abs(r["normalized_entropy"] - nent) <= 1e-12
    assert abs(r["gini"] - gini) <= 1e-12 * max(1.0, abs(gini))
    assert np.allclose(np.asarray(r["probs"], np.float64), g[f"{case}_probs"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("case", ["plain", "expand", "reset", "capped", "collapsed"])
def test_monitor_codebook_against_the_reference_method(case, dev):
    """Golden g13 (the reference's VQGANTrainer.monitor_codebook run on the reference quantizer): the
    module loaded with the stored state, stats.monitor_codebook with the stored settings -- the same
    return value, batch and cumulative statistics, codebook size afterwards and number of re-seeded codes."""
    from vq_vae_amd import EMAVectorQuantizer, stats
    g = np.load(f"{GOLDEN}/g13_monitor_codebook_reference.npz")
    E = g[f"{case}_state_embedding_weight"]
    K, D = E.shape
    m = EMAVectorQuantizer(K, D, 3.0).to(dev)
    sd = {"embedding.weight": E, "ema_cluster_size": g[f"{case}_state_ema_cluster_size"], "ema_w": g[f"{case}_state_ema_w"],
          "usage_count": g[f"{case}_state_usage_count"], "last_reset_epoch": g[f"{case}_state_last_reset_epoch"],
          "last_low_usage_reset_epoch": g[f"{case}_state_last_low_usage_reset_epoch"]}
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.train()
    epoch, disabled, thr, max_size, pct = g[f"{case}_args"]
    rep, calls = {}, []
    out = stats.monitor_codebook(m, torch.from_numpy(g[f"{case}_last_batch_indices"]).to(dev), int(epoch),
                                 max_codebook_size=int(max_size), expansion_threshold=float(thr),
                                 disable_codebook_expansion=bool(disabled), reset_low_usage_percentage=float(pct),
                                 on_expand=calls.append, report=rep, verbose=False)
    assert abs(out - float(g[f"{case}_returned"])) <= 2e-6
    wb, wc = g[f"{case}_batch"], g[f"{case}_cumulative"]
    b, c = rep["batch"], rep["cumulative"]
    assert b["unique"] == int(wb[0]) and abs(b["utilization"] - wb[1]) <= 1e-7
    assert abs(b["entropy"] - wb[2]) <= 2e-6 * wb[2] and abs(b["normalized_entropy"] - wb[3]) <= 2e-6
    assert c["active_size"] == int(wc[0]) and c["utilization"] == wc[1] and c["total_usage"] == wc[4]
    assert abs(c["entropy"] - wc[2]) <= 2e-6 * wc[2] and abs(c["normalized_entropy"] - wc[3]) <= 2e-6
    size = int(g[f"{case}_size_after"])
    assert m.num_embeddings == size and (rep["expanded_to"] == size if size != K else rep["expanded_to"] is None)
    assert len(calls) == (1 if size != K else 0)
    assert [m.embedding.weight.shape[0], m.ema_w.shape[0], m.ema_cluster_size.shape[0], m.usage_count.shape[0]] == \
        list(g[f"{case}_state_shapes_after"])
    assert rep["reset_count"] == int(g[f"{case}_reset_count"])
    assert int(m.last_low_usage_reset_epoch.item()) == int(g[f"{case}_last_reset_epoch_after"])


@pytest.mark.parametrize("case", ["small", "wide", "collapsed", "half"])
def test_count_used_codes_against_the_reference_method(case, dev):
    """Golden g14 (the reference's VQModelTrainer.get_perplexity run on stored latents and codebooks)."""
    from vq_vae_amd import stats
    g = np.load(f"{GOLDEN}/g14_get_perplexity_reference.npz")
    z, E = torch.from_numpy(g[f"{case}_latents"]).to(dev), torch.from_numpy(g[f"{case}_E"]).to(dev)
    if case == "half":
        z = z.half()                                      # the method casts to float32 itself (:59)
    assert stats.count_used_codes(z, E) == int(g[f"{case}_used"])
