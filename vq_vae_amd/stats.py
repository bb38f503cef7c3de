"""Codebook statistics on the GPU: the maths of the reference's analysers applied to true
code indices, without host round trips per code.

* ``CodeUsage`` / ``analyze_indices``: vqvae/codebook_utilization_check.py:83-108 (float64).
* ``batch_code_stats``: vqvae/vqgan_trainer.py:594-612 (fp32).
* ``count_used_codes``: vqvae/trainers/vqtrainer.py:59-75 (search + number of distinct codes).
* ``monitor_codebook``: the trainer's epoch-end check, vqvae/vqgan_trainer.py:577-650.
* ``pairwise_codebook_stats``: the codebook-distance analysis the reference's README.md:26-29
  describes (the reference holds no code for it; the definition is this package's).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import check, load


class CodeUsage:
    """Accumulates a code histogram over batches of indices, then reports the statistics of
    analyze_codebook_utilization (vqvae/codebook_utilization_check.py:83-120)."""

    def __init__(self, num_embeddings: int, device):
        self.K = int(num_embeddings)
        self.device = torch.device(device)
        self.ws = ops.Workspace(self.device)
        self.counts = torch.zeros(self.K, dtype=torch.int32, device=self.device)   # uint32 bit patterns
        self.total = 0

    def update(self, indices: torch.Tensor) -> None:
        ops.histogram(indices, self.K, self.ws, out=self.counts, accumulate=True)
        self.total += indices.numel()

    def all_reduce(self, group=None) -> None:
        """Sum the histogram over the ranks of ``group`` (each rank has seen its own share of the
        indices): one int64 SUM all-reduce of counts[K] and the number of indices
        (vqvae/codebook_utilization_check.py:93-98 and vqvae/vqgan_trainer.py:594-612 over a
        sharded index stream).  Afterwards ``result()`` reports the statistics of the whole stream
        on every rank."""
        from . import distributed as vd
        self.counts, self.total = vd.all_reduce_counts(self.counts, self.total, group)

    def result(self) -> dict:
        lib = load()
        out = torch.empty(5, dtype=torch.float64, device=self.device)
        check(lib.vq_index_stats(ops._ptr(self.counts), self.K, max(self.total, 1), ops._ptr(out), None,
                                 ops._stream(self.device)), "vq_index_stats")
        v = out.cpu().tolist()
        counts = (self.counts.to(torch.int64) & 0xFFFFFFFF).cpu().numpy().astype("float64")
        return {
            "n_embed": self.K,
            "unique_indices": counts.nonzero()[0],
            "utilization": v[1],
            "entropy": v[2],
            "normalized_entropy": v[3],
            "gini": v[4],
            "index_counts": counts,
            "probs": counts / max(self.total, 1),
        }


def analyze_indices(indices: torch.Tensor, num_embeddings: int) -> dict:
    acc = CodeUsage(num_embeddings, indices.device)
    acc.update(indices)
    return acc.result()


def batch_code_stats(indices: torch.Tensor, num_embeddings: int) -> dict:
    """Per-batch utilisation and entropy as monitor_codebook prints them
    (vqvae/vqgan_trainer.py:594-612), one kernel + one 16-byte copy instead of K host syncs."""
    lib = load()
    ws = ops.Workspace(indices.device)
    counts = ops.histogram(indices, num_embeddings, ws)
    out = torch.empty(4, dtype=torch.float32, device=indices.device)
    check(lib.vq_index_stats(ops._ptr(counts), num_embeddings, indices.numel(), None, ops._ptr(out),
                             ops._stream(indices.device)), "vq_index_stats")
    v = out.cpu().tolist()
    return {"unique": int(v[0]), "utilization": v[1], "entropy": v[2], "normalized_entropy": v[3]}


def count_used_codes(latents: torch.Tensor, codebook: torch.Tensor, ws: Optional[ops.Workspace] = None) -> int:
    """Number of distinct nearest codes of ``latents`` ((B, D, H, W) or (N, D)); what
    VQModelTrainer.get_perplexity returns (vqvae/trainers/vqtrainer.py:59-77)."""
    ws = ws or ops.Workspace(latents.device)
    r = ops.search(latents.float(), codebook.float(), ws, want_quantized=False)
    counts = ops.histogram(r["idx32"][: r["N"]], codebook.shape[0], ws)
    return int((counts != 0).sum().item())


def pairwise_codebook_stats(codebook: torch.Tensor, ws: Optional[ops.Workspace] = None, group=None) -> dict:
    """Nearest-other-code distance per code and min / mean / max of all pairwise squared
    distances ||e_i - e_j||^2 (i != j), computed by the MFMA search engine with z := E; the
    K x K matrix is never stored.

    With ``group`` (a torch.distributed process group of more than one rank, every rank holding the
    same codebook) the K x K sweep is sharded by rows: each rank sweeps its rows against all K codes
    (``vq_pairwise_rows``), the per-row nearest neighbours are merged by a MIN all-reduce of packed
    (distance, index) keys over a K-entry buffer, the scalars by one small all-gather; every rank
    returns the full result."""
    lib = load()
    E = ops._f32c(codebook, "codebook")
    K, D = E.shape
    ws = ws or ops.Workspace(E.device)
    image, e2, control = ops.prepare_codebook(E, ws)
    sharded = group is not None and torch.distributed.get_world_size(group) > 1
    if not sharded:
        blocks = lib.vq_search_blocks(K)
        nn_dist = torch.empty(K, dtype=torch.float32, device=E.device)
        nn_idx = torch.empty(K, dtype=torch.int64, device=E.device)
        sum_partial = ws.get("pw_sum", blocks * 8, torch.float64)
        mm_partial = ws.get("pw_mm", (2 * blocks + K) * 4, torch.float32)
        out = torch.empty(3, dtype=torch.float64, device=E.device)
        check(lib.vq_pairwise_stats(ops._ptr(E), ops._ptr(image), ops._ptr(e2), K, D, ops._ptr(nn_dist),
                                    ops._ptr(nn_idx), ops._ptr(sum_partial), ops._ptr(mm_partial), ops._ptr(out),
                                    ops._ptr(control), ops._stream(E.device)), "vq_pairwise_stats")
        v = out.cpu().tolist()
        return {"nn_dist": nn_dist, "nn_idx": nn_idx, "min": v[0], "mean": v[1], "max": v[2]}

    from . import distributed as vd
    rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
    lo, hi = vd.shard_rows(K, rank, world)
    packed = torch.full((K,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=E.device)
    part = pairwise_rows(E, lo, hi - lo, ws, prepared=(image, e2, control))
    if hi > lo:
        packed[lo:hi] = ops.pack_dist_index(part["nn_dist"], part["nn_idx"])
    packed, mn, total, mx = vd.merge_pairwise_shards(packed, part["scalars"], group)
    nn_dist, nn_idx, _ = ops.unpack_dist_index(packed, ws)
    return {"nn_dist": nn_dist, "nn_idx": nn_idx, "min": mn, "mean": total / (K * (K - 1)) if K > 1 else 0.0,
            "max": mx}


def pairwise_rows(codebook: torch.Tensor, row_base: int, n_rows: int, ws: ops.Workspace, prepared=None) -> dict:
    """One rank's share of the row-sharded pairwise sweep: rows [row_base, row_base + n_rows) of the
    codebook against all of it.  Returns nn_dist / nn_idx for those rows (global code numbers) and
    ``scalars`` = float64 device tensor (min, sum, max) over the pairs (i in the range, j != i) --
    (+inf, 0, -inf) for an empty range."""
    lib = load()
    E = ops._f32c(codebook, "codebook")
    K, D = E.shape
    image, e2, control = prepared if prepared is not None else ops.prepare_codebook(E, ws)
    if n_rows <= 0:
        return {"nn_dist": torch.empty(0, dtype=torch.float32, device=E.device),
                "nn_idx": torch.empty(0, dtype=torch.int64, device=E.device),
                "scalars": torch.tensor([float("inf"), 0.0, float("-inf")], dtype=torch.float64, device=E.device)}
    blocks = lib.vq_search_blocks(n_rows)
    nn_dist = torch.empty(n_rows, dtype=torch.float32, device=E.device)
    nn_idx = torch.empty(n_rows, dtype=torch.int64, device=E.device)
    sum_partial = ws.get("pw_sum", blocks * 8, torch.float64)
    mm_partial = ws.get("pw_mm", (2 * blocks + n_rows) * 4, torch.float32)
    out = torch.empty(3, dtype=torch.float64, device=E.device)
    check(lib.vq_pairwise_rows(ops._ptr(E), ops._ptr(image), ops._ptr(e2), K, D, row_base, n_rows, ops._ptr(nn_dist),
                               ops._ptr(nn_idx), ops._ptr(sum_partial), ops._ptr(mm_partial), ops._ptr(out),
                               ops._ptr(control), ops._stream(E.device)), "vq_pairwise_rows")
    return {"nn_dist": nn_dist, "nn_idx": nn_idx, "scalars": out}


@torch.no_grad()
def analyze_codebook_utilization(quantizer, latent_batches, temperature: float = 1.0) -> dict:
    """GPU counterpart of the reference's ``analyze_codebook_utilization``
    (vqvae/codebook_utilization_check.py:64-120) for a stream of latent batches.

    ``latent_batches`` yields the quantizer's inputs ((B, D, H, W) tensors, i.e. what
    ``CustomVQGAN.encode`` returns).  The batches are quantized in eval mode (no EMA update, as
    the reference's ``vqgan.eval()``) and the statistics come from the *index* output rather than
    the one-hot matrix the reference's model forwards under the key "indices".  Returns the same
    keys as the reference: n_embed, unique_indices, utilization, entropy, normalized_entropy,
    gini, index_counts, probs.
    """
    was_training = quantizer.training
    quantizer.eval()
    acc = None
    try:
        for z in latent_batches:
            _, _, (_, idx, _) = quantizer(z, temperature=temperature)
            if acc is None:
                acc = CodeUsage(quantizer.num_embeddings, idx.device)
            acc.update(idx)
    finally:
        quantizer.train(was_training)
    if acc is None:
        raise ValueError("latent_batches was empty")
    return acc.result()


def monitor_codebook(quantizer, last_batch_indices: Optional[torch.Tensor], epoch: int, *,
                     max_codebook_size: int = 512, expansion_threshold: float = 0.8,
                     disable_codebook_expansion: bool = False, reset_low_usage_percentage: float = 0.1,
                     on_expand=None, report: Optional[dict] = None, verbose: bool = True) -> float:
    """The epoch-end codebook check of the reference's trainer (``VQGANTrainer.monitor_codebook``,
    vqvae/vqgan_trainer.py:577-650) for a quantizer on the GPU, with the reference's decisions and
    defaults (:62-63: at most 512 codes, expand above 80 % utilisation):

    * cumulative statistics from ``get_codebook_stats()`` (:580);
    * utilisation and entropy of the last batch's indices (:594-612) -- one histogram kernel and one
      statistics kernel instead of a loop over the codes with a host copy each (:602-603); pass the
      index output of the forward (``perplexity_info[1]``), not the one-hot matrix the reference's model
      hands its trainer;
    * ``expand_codebook`` by 25 % when the cumulative utilisation exceeds the threshold (:622-636);
      ``on_expand(quantizer)`` is called after a successful expansion (the reference rebuilds its
      optimiser there, :636);
    * ``reset_low_usage_codes`` for epochs > 0 (:641-647).

    Returns the cumulative normalised entropy, as the reference does (:650); ``report`` (a dict), when
    given, receives everything that the reference prints."""
    st = quantizer.get_codebook_stats()
    K = quantizer.num_embeddings
    info = {"epoch": epoch, "cumulative": st, "batch": None, "expanded_to": None, "reset_count": 0}
    if last_batch_indices is not None and last_batch_indices.numel() > 0:
        info["batch"] = batch_code_stats(last_batch_indices.reshape(-1), K)
    if (not disable_codebook_expansion and st["utilization"] > expansion_threshold and K < max_codebook_size):
        new_size = min(int(K * 1.25), max_codebook_size)
        if new_size > K and quantizer.expand_codebook(new_size):
            info["expanded_to"] = new_size
            if on_expand is not None:
                on_expand(quantizer)
    if epoch > 0:
        info["reset_count"] = quantizer.reset_low_usage_codes(percentage=reset_low_usage_percentage,
                                                              current_epoch=epoch)
    if verbose:
        print(f"codebook (epoch {epoch}): active {st['active_size']}/{K} ({st['utilization'] * 100:.1f}%), entropy "
              f"{st['entropy']:.4f} (normalised {st['normalized_entropy']:.4f}), total usage {st['total_usage']:.0f}")
        if info["batch"] is not None:
            b = info["batch"]
            print(f"  last batch: {b['unique']}/{K} codes ({b['utilization'] * 100:.1f}%), entropy {b['entropy']:.4f} "
                  f"(normalised {b['normalized_entropy']:.4f})")
        if info["expanded_to"]:
            print(f"  codebook expanded from {K} to {info['expanded_to']}")
        if info["reset_count"]:
            print(f"  {info['reset_count']} low-usage codes re-seeded")
    if report is not None:
        report.update(info)
    return st["normalized_entropy"]


def stats_to_json_dict(results: dict) -> dict:
    """The ``codebook_stats.json`` schema the reference writes with --save_details
    (vqvae/codebook_utilization_check.py:208-217)."""
    return {
        "n_embed": int(results["n_embed"]),
        "unique_indices": [int(v) for v in results["unique_indices"]],
        "utilization": float(results["utilization"]),
        "entropy": float(results["entropy"]),
        "normalized_entropy": float(results["normalized_entropy"]),
        "gini": float(results["gini"]),
        "index_counts": [float(v) for v in results["index_counts"]],
        "probs": [float(v) for v in results["probs"]],
    }


def save_codebook_stats(results: dict, path: str) -> None:
    import json
    with open(path, "w") as f:
        json.dump(stats_to_json_dict(results), f, indent=2)
