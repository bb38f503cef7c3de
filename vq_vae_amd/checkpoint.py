"""Checkpoint helpers for the quantizer's six state tensors.

The reference saves the whole model's ``state_dict`` under ``vqgan_state_dict`` together with a
``model_config`` holding ``vq_num_embed`` / ``vq_embed_dim`` (vqvae/vqgan_trainer.py:409-423), and
its loader recovers (K, D) from the shape of ``quantize.embedding.weight``
(vqvae/codebook_utilization_check.py:33-37) before a strict ``load_state_dict`` (:53).  The
functions here read and write exactly those keys, and convert between one full-codebook state
and per-rank shards of a row-sharded codebook.
"""
from __future__ import annotations

from typing import Dict, List

import torch

QUANTIZER_KEYS = ("ema_cluster_size", "ema_w", "usage_count", "last_reset_epoch", "last_low_usage_reset_epoch",
                  "embedding.weight")
_ROW_SHARDED = ("ema_cluster_size", "ema_w", "usage_count", "embedding.weight")


def quantizer_state(model_state: Dict[str, torch.Tensor], prefix: str = "quantize.") -> Dict[str, torch.Tensor]:
    """Extract the quantizer's six entries from a full model ``state_dict``."""
    out = {}
    for k in QUANTIZER_KEYS:
        full = prefix + k
        if full not in model_state:
            raise KeyError(f"checkpoint lacks '{full}'")
        out[k] = model_state[full]
    return out


def infer_shape(model_state: Dict[str, torch.Tensor], prefix: str = "quantize."):
    """(num_embeddings, embedding_dim) as the reference's loader infers them
    (vqvae/codebook_utilization_check.py:33-37)."""
    w = model_state[prefix + "embedding.weight"]
    return int(w.shape[0]), int(w.shape[1])


def model_config(model_state: Dict[str, torch.Tensor], prefix: str = "quantize.") -> dict:
    """The ``model_config`` entries the reference stores beside the weights
    (vqvae/vqgan_trainer.py:418-421)."""
    K, D = infer_shape(model_state, prefix)
    return {"vq_num_embed": K, "vq_embed_dim": D}


def shard_state(state: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rows [lo, hi) of a full quantizer state for codebook-row sharding; the two epoch counters
    are replicated."""
    from .distributed import shard_rows
    K = state["embedding.weight"].shape[0]
    lo, hi = shard_rows(K, rank, world)
    return {k: (v[lo:hi].clone() if k in _ROW_SHARDED else v.clone()) for k, v in state.items()}


def merge_shards(shards: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Reassemble per-rank shards (in rank order) into the reference's six-tensor layout."""
    out = {}
    for k in QUANTIZER_KEYS:
        if k in _ROW_SHARDED:
            out[k] = torch.cat([s[k] for s in shards], 0)
        else:
            out[k] = shards[0][k].clone()
    return out
