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


def make_stats_reduce(group: Optional[dist.ProcessGroup] = None, uneven_rows: bool = False):
    """Returns the ``stats_reduce`` hook of ``ops.quantize_step`` for data-parallel training.

    Two collectives per step: an int64 all-reduce of the K counts (exact for any batch size)
    and, in training mode, one fp32 all-reduce of dw (K*D*4 bytes: 8 MiB at K=8192, D=256 --
    launch/latency-bound on xGMI, far below a link's bandwidth).
    With ``uneven_rows`` the per-rank row counts are summed too (costs a host sync);
    otherwise every rank is assumed to hold the same number of rows.
    """

    def reduce(counts_i32: torch.Tensor, dw: Optional[torch.Tensor], n_local: int):
        world = dist.get_world_size(group)
        if uneven_rows:
            buf = torch.cat([counts_to_int64(counts_i32),
                             torch.tensor([n_local], dtype=torch.int64, device=counts_i32.device)])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            counts64, n_total = buf[:-1].contiguous(), int(buf[-1].item())
        else:
            counts64 = counts_to_int64(counts_i32)
            dist.all_reduce(counts64, op=dist.ReduceOp.SUM, group=group)
            n_total = n_local * world
        if dw is not None:
            dist.all_reduce(dw, op=dist.ReduceOp.SUM, group=group)
        return counts64, dw, n_total

    return reduce


def shard_rows(K: int, rank: int, world: int):
    """Codebook-row sharding: rank r owns rows [lo, hi) of the K-row codebook."""
    per = (K + world - 1) // world
    lo = min(rank * per, K)
    return lo, min(lo + per, K)


def merge_packed_winners(packed_i64: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """(min-distance, index) all-reduce of the row-sharded search: the packed int64 keys order
    exactly as torch.argmin would order the unsharded distance row (NaN first, then distance,
    then lowest index), so an elementwise MIN over ranks picks the global winner."""
    dist.all_reduce(packed_i64, op=dist.ReduceOp.MIN, group=group)
    return packed_i64
