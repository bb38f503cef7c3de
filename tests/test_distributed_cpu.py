"""Multi-rank host logic on CPU (gloo, world_size 2), plus the algebra of both sharding modes
checked against the unsharded oracle in one process.

Data-parallel mode: per-rank statistics (counts, dw) summed before the EMA update must equal
the statistics of the whole batch.  Codebook-row sharding: per-shard (distance, index) winners
merged by the packed-key MIN must equal the unsharded argmin, ties and NaN included.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from inputs import gauss
from oracle import vq_oracle as vo


def pack_key(d: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """NumPy statement of the packed (distance, index) key of vq_pack_dist_index: NaN first,
    then ascending distance, then ascending index, as a signed int64."""
    d = np.where(d == 0, np.float32(0.0), d).astype(np.float32)      # -0.0 and +0.0 share a key
    u = d.view(np.uint32).astype(np.uint64)
    key = np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)
    key = np.where(np.isnan(d), 0, key).astype(np.uint64)
    packed = (key << np.uint64(32)) | idx.astype(np.uint64)
    return (packed ^ np.uint64(0x8000000000000000)).view(np.int64)


def unpack_index(p: np.ndarray) -> np.ndarray:
    return ((p.view(np.uint64) ^ np.uint64(0x8000000000000000)) & np.uint64(0xFFFFFFFF)).astype(np.int64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, K, D, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vq_vae_amd.distributed import make_stats_reduce, merge_packed_winners, shard_rows
    E = gauss(2, (K, D)) * np.float32(0.7)
    z = gauss(10 + rank, (2, D, 8, 8))                       # this rank's rows
    idx, _ = vo.search(z, E)
    counts = vo.histogram(idx, K)
    dw = vo.embed_sum(z, idx, K)
    # the histogram kernel hands over uint32 bit patterns in an int32 tensor
    c_i32 = torch.from_numpy(counts.astype(np.uint32).view(np.int32).copy())
    red = make_stats_reduce()
    c64, dw_sum, n_total = red(c_i32, torch.from_numpy(dw.copy()), z.size // D)
    red_uneven = make_stats_reduce(uneven_rows=True)
    c64b, _, n_total_b = red_uneven(c_i32, None, z.size // D)
    # codebook-row sharding: this rank searches all rows of both batches against its shard
    zall = np.concatenate([gauss(10 + r, (2, D, 8, 8)) for r in range(world)], 0)
    lo, hi = shard_rows(K, rank, world)
    sidx, sd = vo.search(zall, E[lo:hi], code_base=lo)
    packed = torch.from_numpy(pack_key(sd, sidx).copy())
    merge_packed_winners(packed)
    q.put((rank, c64.numpy(), dw_sum.numpy(), n_total, c64b.numpy(), n_total_b, unpack_index(packed.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_and_row_sharded_over_gloo():
    K, D, world = 128, 32, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, K, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    E = gauss(2, (K, D)) * np.float32(0.7)
    zall = np.concatenate([gauss(10 + r, (2, D, 8, 8)) for r in range(world)], 0)
    idx, _ = vo.search(zall, E)
    counts = vo.histogram(idx, K)
    dw = vo.embed_sum(zall, idx, K)
    for rank, c64, dw_sum, n_total, c64b, n_total_b, merged_idx in res:
        assert np.array_equal(c64, counts) and np.array_equal(c64b, counts)
        assert n_total == zall.size // D and n_total_b == n_total
        assert np.max(np.abs(dw_sum - dw)) <= 1e-5 * np.max(np.abs(dw))
        assert np.array_equal(merged_idx, idx), "row-sharded winners differ from the unsharded argmin"
    # every rank holds identical reduced statistics -> identical updates without a broadcast
    assert np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32))


def test_dp_update_equals_unsharded_update():
    """Algebra of the data-parallel step in one process: 8 row chunks -> per-chunk statistics ->
    sum -> one EMA update == the unsharded oracle step."""
    K, D = 256, 32
    E = gauss(3, (K, D)) * np.float32(0.7)
    z = gauss(4, (16, D, 8, 8))
    full = vo.OracleState.fresh(E)
    vo.forward(z, full, 3.0, 0.95, 1e-5, 1.0, True)
    counts = np.zeros(K, np.int64)
    dw = np.zeros((K, D), np.float32)
    for chunk in np.split(z, 8, axis=0):
        idx, _ = vo.search(chunk, E)
        counts += vo.histogram(idx, K)
        dw += vo.embed_sum(chunk, idx, K)
    st = vo.OracleState.fresh(E)
    vo.ema_update(counts, dw, st.usage_count, st.ema_cluster_size, st.ema_w, st.embedding, 0.95, 1e-5)
    assert np.array_equal(st.usage_count, full.usage_count)
    assert np.max(np.abs(st.ema_cluster_size - full.ema_cluster_size)) <= 1e-6 * np.max(full.ema_cluster_size)
    assert np.max(np.abs(st.embedding - full.embedding)) <= 1e-5 * np.max(np.abs(full.embedding))


def test_row_shard_merge_matches_unsharded_argmin():
    """8 codebook shards, packed-key MIN merge; engineered ties, NaN rows and negative distances."""
    K, D = 512, 16
    E = gauss(5, (K, D))
    E[300] = E[17]                                   # duplicate rows in different shards: lowest index must win
    E[450] = E[17]
    z = gauss(6, (300, D))
    z[:8] = E[17] + np.float32(0.0)                  # exact hits on the duplicated code
    z[9, 3] = np.nan
    idx, dmin = vo.search(z, E)
    merged = None
    for r in range(8):
        lo, hi = r * 64, (r + 1) * 64
        sidx, sd = vo.search(z, E[lo:hi], code_base=lo)
        p = pack_key(sd, sidx)
        merged = p if merged is None else np.minimum(merged, p)
    assert np.array_equal(unpack_index(merged), idx)
    assert (idx[:8] == 17).all()
    # ordering of the key itself
    d = np.array([np.nan, -np.inf, -3.0, -0.0, 0.0, 1e-30, 2.5, np.inf], np.float32)
    keys = pack_key(d, np.zeros(8, np.int64))
    assert keys[0] == keys.min() and np.all(keys[:-1] <= keys[1:]) and keys[3] == keys[4]
    assert pack_key(np.array([1.0], np.float32), np.array([5]))[0] < pack_key(np.array([1.0], np.float32), np.array([6]))[0]


def test_shard_rows_cover_the_codebook():
    from vq_vae_amd.distributed import shard_rows
    for K, world in ((65536, 8), (100, 8), (7, 8)):
        spans = [shard_rows(K, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == K
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _stats_worker(rank, world, port, K, D, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vq_vae_amd.distributed import all_reduce_counts, merge_pairwise_shards, shard_rows
    # (1) the index stream sharded over the ranks: each rank counts its share
    rng = np.random.default_rng(77)
    allidx = rng.zipf(1.3, 50_000) % K
    mine = np.array_split(allidx, world)[rank]
    c = np.bincount(mine, minlength=K).astype(np.uint32)
    if rank == 0:
        c[5] = np.uint32(3_000_000_000)                      # above 2^31: the int32 carrier must not lose it
    counts, total = all_reduce_counts(torch.from_numpy(c.view(np.int32).copy()), mine.size)
    # (2) the K x K pairwise sweep sharded by rows
    E = gauss(31, (K, D))
    E[K - 3] = E[4]                                          # duplicate codes: distance 0, lowest index wins
    lo, hi = shard_rows(K, rank, world)
    nn_dist, nn_idx, _ = vo.pairwise(E)
    Ed = E.astype(np.float64)
    P = ((Ed[lo:hi, None, :] - Ed[None, :, :]) ** 2).sum(-1)
    P[np.arange(hi - lo), np.arange(lo, hi)] = np.nan        # the pair (i, i) is not counted
    packed = np.full(K, np.iinfo(np.int64).max, np.int64)
    packed[lo:hi] = pack_key(nn_dist[lo:hi], nn_idx[lo:hi])
    scal = torch.tensor([np.nanmin(P), np.nansum(P), np.nanmax(P)], dtype=torch.float64)
    merged, mn, tot, mx = merge_pairwise_shards(torch.from_numpy(packed), scal)
    q.put((rank, counts.numpy().view(np.uint32).astype(np.int64), total, unpack_index(merged.numpy()), mn, tot, mx))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_statistics_over_gloo():
    """Index statistics sharded over the stream (int64 SUM of the histograms) and the pairwise
    codebook sweep sharded by rows (packed-key MIN + MIN/SUM/MAX of the scalars), world size 2."""
    K, D, world = 96, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, K, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(77)
    allidx = rng.zipf(1.3, 50_000) % K
    want = np.bincount(allidx, minlength=K).astype(np.int64)
    want[5] += 3_000_000_000 - np.bincount(np.array_split(allidx, world)[0], minlength=K)[5]
    E = gauss(31, (K, D))
    E[K - 3] = E[4]
    _, nn_idx, _ = vo.pairwise(E)
    Ed = E.astype(np.float64)
    P = ((Ed[:, None, :] - Ed[None, :, :]) ** 2).sum(-1)
    off = ~np.eye(K, dtype=bool)
    for rank, counts, total, merged_idx, mn, tot, mx in res:
        assert np.array_equal(counts, want)
        assert total == allidx.size
        assert np.array_equal(merged_idx, nn_idx)
        assert merged_idx[K - 3] == 4 and merged_idx[4] == K - 3
        assert mn == P[off].min() and mx == P[off].max()
        assert abs(tot - P[off].sum()) <= 1e-12 * P[off].sum()


def _packed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vq_vae_amd.distributed import make_stats_reduce
    K, D = 96, 8
    rng = np.random.Generator(np.random.PCG64(77 + rank))
    # counts around the split's boundaries and up to the 32-bit range of the histogram's bins
    c = rng.integers(0, 5000, K).astype(np.uint64)
    c[:6] = [0, 4095, 4096, 4097, (1 << 31) + 5, (1 << 32) - 1 - rank]
    c_i32 = torch.from_numpy(c.astype(np.uint32).view(np.int32).copy())
    dw = torch.from_numpy(rng.standard_normal((K, D)).astype(np.float32))
    n_local = 1000 + 37 * rank
    out = {}
    for name, kw in (("packed", dict(packed=True, uneven_rows=True)), ("plain", dict(packed=False, uneven_rows=True)),
                     ("packed_even", dict(packed=True)), ("plain_even", dict(packed=False))):
        c64, dws, n_total = make_stats_reduce(**kw)(c_i32, dw.clone(), n_local)
        out[name] = (c64.numpy().copy(), dws.numpy().copy(), n_total)
    q.put((rank, c.copy(), out))
    dist.barrier()
    dist.destroy_process_group()


def test_packed_statistics_all_reduce_is_exact():
    """The one-collective form of the data-parallel hook (dw and the counts, split at 12 bits, in one fp32
    buffer) against the two-collective form: identical counts over the whole 32-bit range of a bin,
    identical row totals with uneven batches, bit-identical dw."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (res[0][1] + res[1][1]).astype(np.int64)
    for _rank, _c, out in res:
        for name in ("packed", "plain", "packed_even", "plain_even"):
            assert np.array_equal(out[name][0], want), name
        assert out["packed"][2] == out["plain"][2] == 1000 + 1037
        assert out["packed_even"][2] == out["plain_even"][2]
        assert np.array_equal(out["packed"][1].view(np.uint32), out["plain"][1].view(np.uint32))
        assert np.array_equal(out["packed_even"][1].view(np.uint32), out["plain"][1].view(np.uint32))
    assert np.array_equal(res[0][2]["packed"][1].view(np.uint32), res[1][2]["packed"][1].view(np.uint32))
