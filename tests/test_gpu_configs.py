"""GPU tests at the stated sizes of the benchmark configurations that the other suites only
cover at reduced size: the 10 M-index statistics and the codebook analysis (configuration 3), the
data-parallel module over two ranks (configuration 4) and the K = 65536, D = 256 codebook
sharded by rows into 8 x 8192 (configuration 5)."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import rel_err
from inputs import gauss
from oracle import vq_oracle as vo
from test_gpu_stats import numpy_index_stats, torch_batch_stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("dist_name", ["uniform", "zipf1.1"])
def test_config3_index_statistics_of_10m_indices(dist_name, dev):
    """10 M int64 code indices, K = 8192: the statistics of analyze_codebook_utilization
    (vqvae/codebook_utilization_check.py:86-108, literal NumPy with np.bincount standing in for the
    per-code loop of :94-95) and of monitor_codebook (vqvae/vqgan_trainer.py:594-612, literal torch)."""
    from vq_vae_amd import stats
    K, M = 8192, 10_000_000
    rng = np.random.Generator(np.random.PCG64(303))
    if dist_name == "uniform":
        a = rng.integers(0, K, M, dtype=np.int64)
    else:
        a = np.minimum(rng.zipf(1.1, M) - 1, K - 1).astype(np.int64)
    ref = numpy_index_stats(a, K)
    t = torch.from_numpy(a).to(dev)
    got = stats.analyze_indices(t, K)
    assert np.array_equal(got["index_counts"], ref["index_counts"])          # integers: exact
    assert np.array_equal(got["unique_indices"], ref["unique_indices"])
    assert got["utilization"] == ref["utilization"]
    for key in ("entropy", "normalized_entropy", "gini"):
        assert abs(got[key] - ref[key]) <= 1e-11 * max(1.0, abs(ref[key])), key
    # the same stream in three uneven batches through the accumulator
    acc = stats.CodeUsage(K, dev)
    for part in (t[:3_333_333], t[3_333_333:3_333_334], t[3_333_334:]):
        acc.update(part)
    again = acc.result()
    assert np.array_equal(again["index_counts"], ref["index_counts"]) and again["entropy"] == got["entropy"]
    b = stats.batch_code_stats(t, K)
    tb = torch_batch_stats(a, K)
    assert b["unique"] == tb["unique"]
    assert abs(b["entropy"] - tb["entropy"]) <= 2e-5 * max(1.0, abs(tb["entropy"]))


def _merge_row_ranges(E, bounds, dev):
    """The row-sharded pairwise sweep on one device: every range through vq_pairwise_rows, the
    packed keys merged by vq_packed_min (what the MIN all-reduce does), the scalars by min/sum/max."""
    from vq_vae_amd import ops, stats
    from vq_vae_amd._lib import check, load
    lib = load()
    K = E.shape[0]
    ws = ops.Workspace(dev)
    merged = torch.full((K,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    scal = []
    for lo, hi in bounds:
        part = stats.pairwise_rows(E, lo, hi - lo, ws)
        mine = torch.full((K,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
        if hi > lo:
            mine[lo:hi] = ops.pack_dist_index(part["nn_dist"], part["nn_idx"])
        check(lib.vq_packed_min(ops._ptr(merged), ops._ptr(mine), K, ops._stream(dev)), "vq_packed_min")
        scal.append(part["scalars"].cpu().numpy().copy())
    d, i64, _ = ops.unpack_dist_index(merged, ws)
    scal = np.array(scal)
    return d.cpu().numpy(), i64.cpu().numpy(), scal[:, 0].min(), scal[:, 1].sum(), scal[:, 2].max()


@pytest.mark.parametrize("K,D,parts", [(8192, 256, 8), (65536, 256, 8), (1000, 64, 3)])
def test_config3_pairwise_sweep_sharded_by_rows(K, D, parts, dev):
    """Codebook analysis at K = 8192 (and at K = 65536, the row-sharded codebook): the sweep
    divided into row ranges and merged equals the undivided sweep; both equal the oracle (all rows
    up to K = 8192, the first 256 rows at K = 65536)."""
    from vq_vae_amd import stats
    from vq_vae_amd.distributed import shard_rows
    E = gauss(310 + parts, (K, D)) * np.float32(0.6)
    E[K - 7] = E[11]                                     # duplicated code in another range
    Et = torch.from_numpy(E).to(dev)
    whole = stats.pairwise_codebook_stats(Et)
    bounds = [shard_rows(K, r, parts) for r in range(parts)]
    d, i, mn, tot, mx = _merge_row_ranges(Et, bounds, dev)
    assert np.array_equal(i, whole["nn_idx"].cpu().numpy())
    assert np.array_equal(d, whole["nn_dist"].cpu().numpy())
    assert mn == whole["min"] and mx == whole["max"]
    assert abs(tot / (K * (K - 1)) - whole["mean"]) <= 1e-12 * abs(whole["mean"])
    assert i[K - 7] == 11 and i[11] == K - 7 and abs(d[11]) <= 1e-3        # the expansion rounds: not exactly 0
    if K <= 8192:
        nn_d, nn_i, ref = vo.pairwise(E)
        assert np.array_equal(i, nn_i) and np.array_equal(d.view(np.uint32), nn_d.view(np.uint32))
        assert mn == ref["min"] and mx == ref["max"] and abs(whole["mean"] - ref["mean"]) <= 1e-9 * abs(ref["mean"])
    else:
        si, sd = vo.search(E[:256], E, skip_self=True)
        assert np.array_equal(i[:256], si) and np.array_equal(d[:256].view(np.uint32), sd.view(np.uint32))


def test_code_usage_all_reduce_over_rccl_single_rank(dev):
    """CodeUsage.all_reduce on device tensors through RCCL (one-rank group): the histogram comes back
    unchanged, including a bin above 2^31."""
    import torch.distributed as dist
    from vq_vae_amd import stats
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        K = 8192
        rng = np.random.Generator(np.random.PCG64(11))
        a = rng.integers(0, K, 1_000_000, dtype=np.int64)
        acc = stats.CodeUsage(K, dev)
        acc.update(torch.from_numpy(a).to(dev))
        before = acc.result()
        acc.all_reduce()
        after = acc.result()
        assert np.array_equal(before["index_counts"], after["index_counts"]) and before["entropy"] == after["entropy"]
        acc.counts[7] = -1294967296                              # uint32 3 000 000 000 in the int32 carrier
        acc.all_reduce()
        assert int(acc.counts[7].item()) == -1294967296
    finally:
        if created:
            dist.destroy_process_group()


def test_config5_codebook_of_65536_rows_sharded_8_ways(dev):
    """K = 65536, D = 256 as 8 shards of 8192 rows on one GPU, 64 Ki latents: the sharded step
    (per-shard search with code_base, packed (distance, index) MIN merge, per-shard EMA update)
    equals the unsharded step; indices equal the oracle on a 2048-row sample."""
    from vq_vae_amd import ops
    from vq_vae_amd.distributed import shard_rows, sharded_quantize_step
    K, D, P = 65536, 256, 8
    B, H, W = 64, 32, 32
    N = B * H * W
    E0 = gauss(501, (K, D))
    E0[40000] = E0[123]                                       # duplicate across shards: lowest index must win
    z = gauss(502, (B, D, H, W))
    z[0, :, 0, 0] = E0[123]                                   # a latent that hits the duplicated code exactly
    zt = torch.from_numpy(z).to(dev)
    ws = ops.Workspace(dev)
    full = {"E": torch.from_numpy(E0).to(dev), "ema_w": torch.zeros(K, D, device=dev),
            "cs": torch.zeros(K, device=dev), "usage": torch.zeros(K, device=dev)}
    ref = ops.quantize_step(zt, full["E"], full["ema_w"], full["cs"], full["usage"], beta=3.0, decay=0.95,
                            epsilon=1e-5, temperature=1.0, training=True, ws=ws)
    shards = []
    for r in range(P):
        lo, hi = shard_rows(K, r, P)
        assert hi - lo == 8192
        shards.append({"lo": lo, "E": torch.from_numpy(E0[lo:hi].copy()).to(dev),
                       "ema_w": torch.zeros(hi - lo, D, device=dev), "cs": torch.zeros(hi - lo, device=dev),
                       "usage": torch.zeros(hi - lo, device=dev), "ws": ops.Workspace(dev)})
    packs = []
    for s in shards:
        r_ = ops.search(zt, s["E"], s["ws"], code_base=s["lo"], want_quantized=False, want_dmin=True)
        packs.append(ops.pack_dist_index(r_["dmin"], r_["indices"]))
    merged = torch.stack(packs).min(0).values
    del packs
    E_full = torch.cat([s["E"] for s in shards], 0).clone()
    n_total = torch.tensor(float(np.float32(1 - 0.95)) * float(N), dtype=torch.float64, device=dev)
    for s in shards:
        o = sharded_quantize_step(zt, s["E"], s["ema_w"], s["cs"], s["usage"], K_total=K, code_base=s["lo"],
                                  beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=True,
                                  ws=s["ws"], merge=lambda p: merged.clone(),
                                  gather_codebook=lambda e: E_full, sum_scalar=lambda n: n_total)
        assert torch.equal(o.indices, ref.indices)
        assert torch.equal(o.quantized, ref.quantized)
        assert abs(o.loss.item() - ref.loss.item()) <= 1e-6 * abs(ref.loss.item())
        assert abs(o.perplexity.item() - ref.perplexity.item()) <= 1e-6 * ref.perplexity.item()
    assert rel_err(torch.cat([s["E"] for s in shards]).cpu().numpy(), full["E"].cpu().numpy()) <= 1e-5
    assert rel_err(torch.cat([s["cs"] for s in shards]).cpu().numpy(), full["cs"].cpu().numpy()) <= 1e-5
    assert torch.equal(torch.cat([s["usage"] for s in shards]), full["usage"])
    # (the row slabs of the EMA sums are planned per call, so the fp32 addition of slab totals groups
    # the rows differently for a shard and for the whole codebook: close, not bit-equal)
    assert rel_err(torch.cat([s["ema_w"] for s in shards]).cpu().numpy(), full["ema_w"].cpu().numpy()) <= 1e-6
    idx = ref.indices.cpu().numpy()
    assert idx[0] == 123
    # oracle on a row sample: the first two images' first 1024 rows each (NCHW rows b*HW + hw)
    sample = z[:2].copy()
    si, _ = vo.search(sample, E0)
    assert np.array_equal(idx[: 2 * H * W], si)
    # size-independent properties on all 64 Ki rows
    q = ref.quantized.permute(0, 2, 3, 1).reshape(N, D)
    zz = zt.permute(0, 2, 3, 1).reshape(N, D)
    assert torch.equal(q, zz + (torch.from_numpy(E0).to(dev)[ref.indices] - zz))
    counts = (ref.counts.to(torch.int64) & 0xFFFFFFFF)
    assert int(counts.sum()) == N and torch.equal(counts, torch.bincount(ref.indices, minlength=K))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_module_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vq_vae_amd import EMAVectorQuantizer
        dev = torch.device("cuda:0")                       # both ranks share the one GPU of the box
        K, D = 512, 64
        m = EMAVectorQuantizer(K, D, 3.0, decay=0.95).to(dev)
        with torch.no_grad():
            m.embedding.weight.copy_(torch.from_numpy(gauss(700, (K, D)) * np.float32(0.5)))
        m.enable_data_parallel()
        m.train()
        perps = []
        for step in range(3):
            z = torch.from_numpy(gauss(710 + 10 * step + rank, (4, D, 16, 16))).to(dev)
            _, _, (perp, idx, _) = m(z)
            perps.append(float(perp.item()))
        torch.cuda.synchronize()
        q.put((rank, {k: v.cpu().numpy() for k, v in m.state_dict().items()}, perps))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_config4_data_parallel_module_two_ranks(dev):
    """The module with enable_data_parallel() on two ranks (gloo, both on this GPU), three training
    steps on different batches per rank: the replicas' six state tensors stay bit-identical, and
    equal the oracle's trajectory on the concatenated batches."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_module_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=110) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, p0), (_, s1, p1) = res
    for k in s0:
        assert np.array_equal(s0[k].view(np.uint8), s1[k].view(np.uint8)), k
    assert p0 == p1
    K, D = 512, 64
    st = vo.OracleState.fresh(gauss(700, (K, D)) * np.float32(0.5))
    for step in range(3):
        zcat = np.concatenate([gauss(710 + 10 * step + r, (4, D, 16, 16)) for r in range(world)], 0)
        out = vo.forward(zcat, st, 3.0, 0.95, 1e-5, 1.0, True)
        assert abs(p0[step] - out["perplexity"]) <= 1e-5 * out["perplexity"]
    assert rel_err(s0["embedding.weight"], st.embedding) <= 1e-5
    assert rel_err(s0["ema_w"], st.ema_w) <= 1e-5
    assert rel_err(s0["ema_cluster_size"], st.ema_cluster_size) <= 1e-5
    assert np.array_equal(s0["usage_count"], st.usage_count)


def _dp_boundary_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vq_vae_amd import QuantizerBoundary
        dev = torch.device("cuda:0")                       # both ranks share the one GPU of the box
        K, D = 8192, 256
        m = QuantizerBoundary(D, K, D, ema_decay=0.95, commitment_loss_beta=3.0).to(dev)
        with torch.no_grad():
            m.quantize.embedding.weight.copy_(torch.from_numpy(gauss(900, (K, D))))
            m.quant_conv.weight.copy_(torch.from_numpy(gauss(901, (D, D, 1, 1)) * np.float32(0.06)))
            m.quant_conv.bias.copy_(torch.from_numpy(gauss(902, (D,)) * np.float32(0.1)))
            m.post_quant_conv.weight.copy_(torch.from_numpy(gauss(903, (D, D, 1, 1)) * np.float32(0.06)))
            m.post_quant_conv.bias.copy_(torch.from_numpy(gauss(904, (D,)) * np.float32(0.1)))
        m.quantize.enable_data_parallel()
        m.train()
        conv_params = [m.quant_conv.weight, m.quant_conv.bias, m.post_quant_conv.weight, m.post_quant_conv.bias]

        def step(h_np, w_np):
            m.zero_grad(set_to_none=True)
            h = torch.from_numpy(h_np).to(dev).requires_grad_(True)
            out, loss, (perp, idx, _) = m(h)
            ((out * torch.from_numpy(w_np).to(dev)).sum() + loss).backward()
            for p in conv_params:                      # what DistributedDataParallel does: average over the ranks
                g = p.grad.cpu()
                dist.all_reduce(g)
                p.grad.copy_(g / world)
            torch.cuda.synchronize()
            return out, loss, perp, idx, h.grad

        # the per-rank batch of 8 images, checked against the oracle on the 16 images of both ranks
        out, loss, perp, idx, gh = step(gauss(910 + rank, (8, D, 32, 32)), gauss(920 + rank, (8, D, 32, 32)))
        first = {"loss": float(loss), "perp": float(perp), "idx": idx.cpu().numpy(), "out": out.detach().cpu().numpy(),
                 "grad_h": gh.cpu().numpy(), "grads": [p.grad.cpu().numpy() for p in conv_params],
                 "state": {k: v.cpu().numpy() for k, v in m.quantize.state_dict().items()}}
        # configuration 4's stated per-GPU batch: 64 images of 32 x 32 latents
        out, loss, perp, idx, gh = step(gauss(930 + rank, (64, D, 32, 32)), gauss(940 + rank, (64, D, 32, 32)))
        counts = torch.bincount(idx.view(-1), minlength=K).cpu().numpy()
        second = {"loss": float(loss), "perp": float(perp), "counts": counts,
                  "grads": [p.grad.cpu().numpy() for p in conv_params],
                  "state": {k: v.cpu().numpy() for k, v in m.quantize.state_dict().items()}}
        q.put((rank, first, second))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_config4_boundary_segment_data_parallel_at_its_stated_shape(dev):
    """Configuration 4 through the whole segment: quant_conv -> quantizer (K = 8192, D = 256) -> post_quant_conv
    (vqvae/custom_vqgan.py:76-114 between encoder and decoder) on two ranks, EMA statistics all-reduced, the
    convolutions' gradients averaged over the ranks as DistributedDataParallel would.  8 images per rank against
    the oracle's composition on the 16 images of both ranks; then one step at the stated 64 images per rank.
    The replicas must stay bit-identical throughout."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_boundary_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=280) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, a1, a2), (_, b1, b2) = res
    # replicas: state and averaged gradients bit for bit, after both steps
    for x, y in ((a1, b1), (a2, b2)):
        for k in x["state"]:
            assert np.array_equal(x["state"][k].view(np.uint8), y["state"][k].view(np.uint8)), k
        for gx, gy in zip(x["grads"], y["grads"]):
            assert np.array_equal(gx.view(np.uint8), gy.view(np.uint8))
        assert x["perp"] == y["perp"]
    # the oracle on the concatenated batch of step one
    K, D = 8192, 256
    Wq, bq = gauss(901, (D, D, 1, 1)) * np.float32(0.06), gauss(902, (D,)) * np.float32(0.1)
    Wp, bp = gauss(903, (D, D, 1, 1)) * np.float32(0.06), gauss(904, (D,)) * np.float32(0.1)
    h = np.concatenate([gauss(910 + r, (8, D, 32, 32)) for r in range(world)], 0)
    w = np.concatenate([gauss(920 + r, (8, D, 32, 32)) for r in range(world)], 0)
    st = vo.OracleState.fresh(gauss(900, (K, D)))
    # averaged over two ranks: d/dtheta of (S_0 + S_1) / 2 + (L_0 + L_1) / 2, and (L_0 + L_1) / 2 is the mean over all rows
    r = vo.boundary(h, Wq, bq, Wp, bp, st, 3.0, 0.95, 1e-5, 1.0, True, grad_out=w / np.float32(world), grad_loss=1.0)
    idx = np.concatenate([a1["idx"].reshape(-1), b1["idx"].reshape(-1)])
    assert np.array_equal(idx, r["indices"].reshape(-1))
    assert rel_err(np.concatenate([a1["out"], b1["out"]], 0), r["out"]) <= 1e-5
    assert abs((a1["loss"] + b1["loss"]) / 2 - r["loss"]) <= 1e-5 * abs(r["loss"])
    assert abs(a1["perp"] - r["perplexity"]) <= 1e-5 * r["perplexity"]
    for got, want in zip(a1["grads"], (r["grad_Wq"], r["grad_bq"], r["grad_Wp"], r["grad_bp"])):
        assert rel_err(got, want) <= 1e-5
    # each rank's input gradient: its own loss has weight 1, not 1/2, and the gradient is linear in the two weights
    gh = np.concatenate([a1["grad_h"], b1["grad_h"]], 0)
    assert rel_err(gh, np.float32(world) * r["grad_h"]) <= 1e-5
    assert rel_err(a1["state"]["embedding.weight"], st.embedding) <= 1e-5
    assert rel_err(a1["state"]["ema_w"], st.ema_w) <= 1e-5
    assert rel_err(a1["state"]["ema_cluster_size"], st.ema_cluster_size) <= 1e-5
    assert np.array_equal(a1["state"]["usage_count"], st.usage_count)
    # step two (64 images per rank): every row counted once on each rank, usage grew by all 128 Ki rows of both ranks
    assert int(a2["counts"].sum()) == 64 * 1024 and int(b2["counts"].sum()) == 64 * 1024
    grown = a2["state"]["usage_count"].astype(np.float64).sum() - a1["state"]["usage_count"].astype(np.float64).sum()
    assert grown == 2 * 64 * 1024
    assert np.array_equal(a2["state"]["usage_count"] - a1["state"]["usage_count"], (a2["counts"] + b2["counts"]).astype(np.float32))
    assert np.isfinite(a2["loss"]) and np.isfinite(a2["perp"]) and all(np.isfinite(g).all() for g in a2["grads"])
