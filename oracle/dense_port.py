"""Dense one-hot PyTorch formulation of the quantizer forward.  TEST INFRASTRUCTURE ONLY.

The reference computes its forward with dense N x K matrices: a distance matrix, a one-hot
matrix, and two matrix products with that one-hot (vqvae/ema_vector_quantizer.py:42-46, :54-58,
:84).  This module states the same sequence of tensor operations as a function over an explicit
state, so that the benchmark can time "what the reference's formulation costs on the host CPU"
on a machine the reference itself cannot travel to, and so that tests have a second,
independently written checker beside the C oracle.  It is validated against the golden vectors
in tests/test_oracle_golden.py::test_dense_port_matches_golden.
"""
from __future__ import annotations

import torch


def new_state(E: torch.Tensor) -> dict:
    K, D = E.shape
    return {"E": E.clone(), "cs": torch.zeros(K), "ema_w": torch.zeros(K, D), "usage": torch.zeros(K)}


@torch.no_grad()
def dense_forward(z_nchw: torch.Tensor, st: dict, beta: float, decay: float = 0.95, eps: float = 1e-5,
                  temperature: float = 1.0, training: bool = True):
    """One forward over a (B, D, H, W) batch with dense N x K intermediates."""
    E = st["E"]
    K, D = E.shape
    x = z_nchw.permute(0, 2, 3, 1).contiguous()          # :35
    flat = x.view(-1, D)                                  # :36
    N = flat.shape[0]
    dist = (flat.pow(2).sum(1, keepdim=True) + E.pow(2).sum(1)) - 2 * (flat @ E.t())   # :42-46
    if temperature != 1.0:
        dist = dist / temperature                         # :49-50
    idx = dist.argmin(1, keepdim=True)                    # :53
    onehot = torch.zeros(N, K).scatter_(1, idx, 1)        # :54-55
    q = (onehot @ E).view(x.shape)                        # :58
    col = onehot.sum(0)
    if training:
        st["usage"] = st["usage"] + col                   # :65-67
        cs = st["cs"] * decay + (1 - decay) * col         # :71-72
        n = cs.sum()                                      # :75
        cs = (cs + eps) / (n + K * eps) * n               # :76-80
        st["cs"] = cs
        st["ema_w"] = st["ema_w"] * decay + (1 - decay) * (onehot.t() @ flat)   # :84-85
        st["E"] = st["ema_w"] / cs.unsqueeze(1)           # :89
    loss = beta * torch.mean((q - x) ** 2)                # :92
    q_st = (x + (q - x)).permute(0, 3, 1, 2).contiguous() # :95, :98
    p = onehot.mean(0)
    perplexity = torch.exp(-(p * torch.log(p + 1e-10)).sum())   # :105-106
    return q_st, loss, perplexity, idx
