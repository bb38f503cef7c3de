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


def load_reference_checkpoint(path: str, map_location="cpu") -> dict:
    """Read a checkpoint file written by the reference's trainer (full, ``_best`` or ``_best_light``
    variant, vqvae/vqgan_trainer.py:409-445) with a loader that executes nothing from the file
    (``torch.load(weights_only=True)``).  Returns the quantizer's six tensors, the (K, D) inferred from
    them, and the bookkeeping entries the reference stores beside the weights."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if "vqgan_state_dict" not in ckpt:
        raise KeyError("not a VQGAN trainer checkpoint: no 'vqgan_state_dict'")
    model_state = ckpt["vqgan_state_dict"]
    K, D = infer_shape(model_state)
    cfg = ckpt.get("model_config", {})
    if cfg and int(cfg.get("vq_num_embed", K)) != K:
        # the reference refuses such a file too (vqvae/vqgan_trainer.py:484-487)
        raise ValueError(f"model_config says {cfg['vq_num_embed']} codes, the weights hold {K}")
    return {"quantizer_state": quantizer_state(model_state), "num_embeddings": K, "embedding_dim": D,
            "model_state": model_state, "model_config": cfg, "epoch": ckpt.get("epoch", 0),
            "best_val_loss": ckpt.get("best_val_loss", 0.0), "codebook_stats": ckpt.get("codebook_stats", [])}


def light_checkpoint(model_state: Dict[str, torch.Tensor], epoch: int, best_val_loss: float = 0.0,
                     codebook_stats=None, extra: dict = None,
                     discriminator_state: Dict[str, torch.Tensor] = None,
                     g_scheduler_state: dict = None, d_scheduler_state: dict = None) -> dict:
    """The dict of the reference's light checkpoint variant: the trainer's checkpoint
    (vqvae/vqgan_trainer.py:409-423) without the entries whose key contains 'optimizer' or 'scaler'
    (:442) -- so the discriminator's weights and the schedulers' states stay in it.

    Readers: vqvae/codebook_utilization_check.py:18-53 needs only 'vqgan_state_dict'.  The trainer's
    own loader (vqvae/vqgan_trainer.py:474-526) also indexes 'discriminator_state_dict' (:492) and gives
    up without it, so pass ``discriminator_state`` for a file the trainer is to resume from
    (with ``skip_optimizer``; the optimizer entries are what the light variant leaves out);
    ``g_scheduler_state`` / ``d_scheduler_state`` are stored under the keys it looks for (:501-504)."""
    out = {"epoch": int(epoch), "vqgan_state_dict": {k: v.detach().cpu() for k, v in model_state.items()},
           "best_val_loss": float(best_val_loss), "codebook_stats": list(codebook_stats or []),
           "model_config": model_config(model_state)}
    if discriminator_state is not None:
        out["discriminator_state_dict"] = {k: v.detach().cpu() for k, v in discriminator_state.items()}
    if g_scheduler_state is not None:
        out["g_scheduler_state_dict"] = g_scheduler_state
    if d_scheduler_state is not None:
        out["d_scheduler_state_dict"] = d_scheduler_state
    out.update(extra or {})
    return out
