"""Multi-GPU modes of the quantizer: one process per GPU, torch.distributed over RCCL (xGMI).

Data-parallel over rows (the reference has no multi-device code on this path; the exchange
sits between the statistics of vqvae/ema_vector_quantizer.py:65/:84 and the update of
:71-89): every rank holds the full codebook and EMA state, searches its own rows, and the
EMA statistics -- counts[K] and dw[K, D] -- are summed over ranks before the update, so every
replica applies the same update and stays bit-identical without a broadcast.

The collectives work on whatever device the tensors live on, so the host logic is testable
with the gloo backend on CPU tensors.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def counts_to_int64(counts_i32: torch.Tensor) -> torch.Tensor:
    """The histogram kernel stores uint32 counts in an int32 tensor; widen without sign loss."""
    return counts_i32.to(torch.int64) & 0xFFFFFFFF


def _all_reduce(t: torch.Tensor, op, group) -> None:
    """dist.all_reduce in place.  RCCL reduces device tensors directly; the gloo backend (CPU tests and
    rehearsals of the multi-rank path on a box with fewer GPUs) gets a host copy, because its own
    handling of device tensors stalls when several ranks share one GPU."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def make_stats_reduce(group: Optional[dist.ProcessGroup] = None, uneven_rows: bool = False, packed: bool = True):
    """Returns the ``stats_reduce`` hook of ``ops.quantize_step`` for data-parallel training.

    Training mode, ``packed`` (the default): ONE fp32 all-reduce per step over [dw (K*D) | counts low 12
    bits (K) | counts high bits (K)] -- 8.06 MiB at K=8192, D=256, latency-bound on xGMI, so one
    collective instead of two is what counts for a per-GPU batch whose whole step takes half a
    millisecond.  The counts stay exact: a count c travels as c mod 4096 and c div 4096, both sums are
    integers below 2^24 for up to 4096 ranks of fewer than 2^24 rows each, and fp32 adds such integers
    exactly.  Which form runs must not depend on anything that differs between ranks (ranks that
    chose differently would issue different collectives and hang): it is decided from the world
    size alone, and a rank holding 2^24 rows or more raises instead of switching.
    Otherwise (``packed=False``, more than 4096 ranks, or eval mode, where there is no dw): an int64
    all-reduce of the K counts and, in training mode, an fp32 all-reduce of dw.
    With ``uneven_rows`` the per-rank row counts are summed too (costs a host sync);
    otherwise every rank is assumed to hold the same number of rows.
    """
    cache = {}

    def reduce(counts_i32: torch.Tensor, dw: Optional[torch.Tensor], n_local: int):
        world = dist.get_world_size(group)
        if packed and dw is not None and world <= 4096:
            if n_local >= 1 << 24:
                raise ValueError(f"{n_local} rows on one rank: the packed statistics all-reduce carries counts "
                                 "exactly below 2^24 rows per rank; split the batch or use packed=False")
            K = counts_i32.numel()
            kd = dw.numel()
            extra = 2 if uneven_rows else 0
            key = (dw.device, kd, K, extra)
            buf = cache.get(key)
            if buf is None:
                cache.clear()
                buf = cache[key] = torch.empty(kd + 2 * K + extra, dtype=torch.float32, device=dw.device)
            c = counts_to_int64(counts_i32)
            buf[:kd].copy_(dw.view(-1))
            buf[kd:kd + K].copy_(c & 4095)
            buf[kd + K:kd + 2 * K].copy_(c >> 12)
            if uneven_rows:
                buf[kd + 2 * K] = float(n_local & 4095)
                buf[kd + 2 * K + 1] = float(n_local >> 12)
            _all_reduce(buf, dist.ReduceOp.SUM, group)
            dw.view(-1).copy_(buf[:kd])
            counts64 = buf[kd:kd + K].to(torch.int64) + (buf[kd + K:kd + 2 * K].to(torch.int64) << 12)
            if uneven_rows:
                tail = buf[kd + 2 * K:].to(torch.int64).cpu()
                n_total = int(tail[0]) + (int(tail[1]) << 12)
            else:
                n_total = n_local * world
            return counts64, dw, n_total
        if uneven_rows:
            buf = torch.cat([counts_to_int64(counts_i32),
                             torch.tensor([n_local], dtype=torch.int64, device=counts_i32.device)])
            _all_reduce(buf, dist.ReduceOp.SUM, group)
            counts64, n_total = buf[:-1].contiguous(), int(buf[-1].item())
        else:
            counts64 = counts_to_int64(counts_i32)
            _all_reduce(counts64, dist.ReduceOp.SUM, group)
            n_total = n_local * world
        if dw is not None:
            _all_reduce(dw, dist.ReduceOp.SUM, group)
        return counts64, dw, n_total

    def reduce_packed(buf: torch.Tensor):
        """SUM all-reduce of a buffer the library packed on the device (vq_pack_stats).  Over RCCL the
        collective is issued asynchronously, on the backend's own stream, and the function returned makes
        the caller's current stream wait for it: whatever the caller enqueues in between runs beside the
        exchange.  (gloo, with ranks sharing a GPU: a blocking host round trip.)"""
        if buf.is_cuda and dist.get_backend(group) != "gloo":
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
            return work.wait
        _all_reduce(buf, dist.ReduceOp.SUM, group)
        return lambda: None

    reduce.reduce_packed = reduce_packed
    reduce.world_size = lambda: dist.get_world_size(group)
    return reduce


def all_reduce_counts(counts_i32: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None):
    """Statistics sharded over the index stream: every rank holds the histogram of its share
    (uint32 bit patterns in an int32 tensor) and the number of indices it counted.  One int64 SUM
    all-reduce of [counts ‖ total]; returns (counts in the same uint32-in-int32 form, total).
    Raises if a summed bin no longer fits 32 bits (the statistics kernel reads uint32 bins)."""
    buf = torch.cat([counts_to_int64(counts_i32),
                     torch.tensor([int(total)], dtype=torch.int64, device=counts_i32.device)])
    _all_reduce(buf, dist.ReduceOp.SUM, group)
    counts64, n_total = buf[:-1], int(buf[-1].item())
    if int(counts64.max().item()) >= 1 << 32:
        raise OverflowError("a summed histogram bin exceeds 32 bits")
    wrapped = torch.where(counts64 >= 1 << 31, counts64 - (1 << 32), counts64).to(torch.int32)
    return wrapped.contiguous(), n_total


def merge_pairwise_shards(packed_i64: torch.Tensor, scalars_f64: torch.Tensor,
                          group: Optional[dist.ProcessGroup] = None):
    """Merge of the row-sharded pairwise codebook sweep.  ``packed_i64`` [K]: packed (distance,
    index) keys of the nearest other code for the rows this rank swept, INT64_MAX elsewhere -- one
    MIN all-reduce fills every entry.  ``scalars_f64`` (min, sum, max) of this rank's pairs: one
    all-gather, combined as MIN / SUM / MAX (NaN from any rank propagates).
    Returns (packed, min, sum, max)."""
    _all_reduce(packed_i64, dist.ReduceOp.MIN, group)
    world = dist.get_world_size(group)
    mine = scalars_f64.detach().to(torch.float64).reshape(3)
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        mine = mine.cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine.contiguous(), group=group)
    allv = torch.stack(parts).cpu()
    mn, total, mx = allv[:, 0], allv[:, 1], allv[:, 2]
    nan = bool(torch.isnan(allv).any())
    f = float("nan")
    return (packed_i64, f if nan else float(mn.min()), f if nan else float(total.sum()), f if nan else float(mx.max()))


def shard_rows(K: int, rank: int, world: int):
    """Codebook-row sharding: rank r owns rows [lo, hi) of the K-row codebook."""
    per = (K + world - 1) // world
    lo = min(rank * per, K)
    return lo, min(lo + per, K)


def merge_packed_winners(packed_i64: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """(min-distance, index) all-reduce of the row-sharded search: the packed int64 keys order
    exactly as torch.argmin would order the unsharded distance row (NaN first, then distance,
    then lowest index), so an elementwise MIN over ranks picks the global winner."""
    _all_reduce(packed_i64, dist.ReduceOp.MIN, group)
    return packed_i64


def sharded_quantize_step(z: torch.Tensor, E_shard: torch.Tensor, ema_w_shard, cluster_size_shard, usage_shard, *,
                          K_total: int, code_base: int, beta: float, decay: float, epsilon: float,
                          temperature: float, training: bool, ws, group: Optional[dist.ProcessGroup] = None,
                          merge=None, gather_codebook=None, sum_scalar=None, filtered: bool = False):
    """Forward of a quantizer whose codebook rows are sharded over the ranks of ``group``
    (large codebooks, e.g. K = 65536 as 8 x 8192 rows); every rank sees the same rows ``z``.

    Exchange steps: (1) an int64 MIN all-reduce of the packed (distance, index) winners, N*8
    bytes; (2) an all-gather of the codebook shards for the gather of the quantized output,
    K*D*4 bytes; (3) one scalar SUM for the Laplace normaliser n = sum(cluster_size).  Each rank
    then updates only its own rows.  ``merge`` / ``gather_codebook`` / ``sum_scalar`` default to
    the torch.distributed collectives and can be replaced (single-process tests).  ``filtered``
    runs each shard's search through the filtered kernels (same winners and distances).
    """
    from . import ops
    from ._lib import check, load
    lib = load()
    K_loc, D = E_shard.shape
    r = ops.search(z, E_shard, ws, temperature=temperature, code_base=code_base, want_quantized=False,
                   want_dmin=True, filtered=filtered)
    packed = ops.pack_dist_index(r["dmin"], r["indices"])
    packed = merge(packed) if merge is not None else merge_packed_winners(packed, group)
    _, idx64, idx32 = ops.unpack_dist_index(packed, ws, want_dmin=False)
    N = r["N"]
    # quantized output: needs the winners' rows, wherever they live
    if gather_codebook is not None:
        E_full = gather_codebook(E_shard)
    else:
        world = dist.get_world_size(group)
        parts = [torch.empty_like(E_shard) for _ in range(world)]
        dist.all_gather(parts, E_shard.contiguous(), group=group)
        E_full = torch.cat(parts, 0)[:K_total].contiguous()
    q, lp, n_partial = ops.gather_st(z, E_full, idx64, ws)
    counts_full = ops.histogram(idx32[:N], K_total, ws)          # every rank has all winners
    loss, perp, _ = ops.step_scalars(lp, n_partial, N, D, beta, counts_full, K_total, z.device)
    if training:
        slabs, _ = ops.embed_sums(ops._f32c(z, "latents"), idx32, K_loc, ws, code_base=code_base, zmax_valid=True)
        counts_loc = counts_full[code_base: code_base + K_loc].contiguous()
        d, omd, eps, keps = ops.ema_scalars(K_total, decay, epsilon)
        # cluster sizes: local EMA, global Laplace normaliser
        c = (counts_loc.to(torch.int64) & 0xFFFFFFFF).float()
        usage_shard += c
        cluster_size_shard.mul_(d).add_(c * omd)
        n = cluster_size_shard.double().sum()
        n = sum_scalar(n) if sum_scalar is not None else _all_reduce_sum(n, group)
        n = n.float()
        cluster_size_shard.copy_((cluster_size_shard + eps) / (n + keps) * n)
        check(lib.vq_ema_codebook_update(ops._ptr(slabs), slabs.shape[0], K_loc, D, d, omd,
                                         ops._ptr(cluster_size_shard), ops._ptr(ema_w_shard), ops._ptr(E_shard),
                                         ops._stream(z.device)), "vq_ema_codebook_update")
    return ops.StepResult(q, loss, perp, idx64, counts_full, None)


def _all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    t = t.clone()
    _all_reduce(t, dist.ReduceOp.SUM, group)
    return t
