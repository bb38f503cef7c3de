"""Near-tie census: how many rows does this library's search assign differently from the reference's
dense formulation, and is every such row a near-tie?

The reference computes all N x K distances with one sgemm (vqvae/ema_vector_quantizer.py:42-46) and takes
torch.argmin (:53).  The order in which a BLAS accumulates z.e is its own, so on rows whose two best
codes are within a few fp32 roundings of each other it may name the other code.  The contract of
this build: such rows are counted and each is shown to be a near-tie --

    |d_a - d_b|  <=  c * eps_fp32 * (||z||^2 + max(||e_a||^2, ||e_b||^2)) / max(|T|, 1),   c = 8,

with both distances recomputed in float64 -- and nothing else may differ.  Measurement / test
infrastructure, not part of the product: works on torch tensors of any device.
"""
from __future__ import annotations

import torch

NEAR_TIE_C = 8.0
EPS32 = float(torch.finfo(torch.float32).eps)


def rows_of(z: torch.Tensor) -> torch.Tensor:
    """(B, D, H, W) -> (N, D) rows in the reference's order (vqvae/ema_vector_quantizer.py:35-36)."""
    if z.dim() == 2:
        return z
    return z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])


@torch.no_grad()
def dense_indices(z: torch.Tensor, E: torch.Tensor, temperature: float = 1.0, chunk: int = 65536) -> torch.Tensor:
    """The reference's formulation of the search, op for op (:42-53), in chunks of rows so that the
    N x K matrix fits: ||z||^2 + ||e||^2 - 2 z E^T by torch.matmul (MKL on the host, rocBLAS on the GPU),
    optional division by the temperature, torch.argmin."""
    flat = rows_of(z)
    e2 = torch.sum(E ** 2, dim=1)
    out = []
    for lo in range(0, flat.shape[0], chunk):
        f = flat[lo: lo + chunk].contiguous()
        dist = torch.sum(f ** 2, dim=1, keepdim=True) + e2 - 2 * torch.matmul(f, E.t())
        if temperature != 1.0:
            dist = dist / temperature
        out.append(torch.argmin(dist, dim=1))
        del dist
    return torch.cat(out)


@torch.no_grad()
def census(z: torch.Tensor, E: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor,
           temperature: float = 1.0) -> dict:
    """Compare two index vectors over the same rows; classify every differing row in float64."""
    flat = rows_of(z)
    a = idx_a.reshape(-1).to(flat.device, torch.int64)
    b = idx_b.reshape(-1).to(flat.device, torch.int64)
    bad = torch.nonzero(a != b).reshape(-1)
    n_bad = int(bad.numel())
    res = {"rows": int(a.numel()), "mismatches": n_bad, "near_ties": 0, "not_near_ties": 0,
           "criterion": f"|d_a - d_b| <= {NEAR_TIE_C:g} * eps_fp32 * (|z|^2 + max|e|^2) / max(|T|, 1), float64 distances",
           "largest_gap_over_bound": 0.0}
    if n_bad == 0:
        return res
    zf = flat[bad].double()
    ea, eb = E[a[bad]].double(), E[b[bad]].double()
    da = torch.sum((zf - ea) ** 2, dim=1)
    db = torch.sum((zf - eb) ** 2, dim=1)
    scale = torch.sum(zf ** 2, dim=1) + torch.maximum(torch.sum(ea ** 2, dim=1), torch.sum(eb ** 2, dim=1))
    bound = NEAR_TIE_C * EPS32 * scale / max(abs(float(temperature)), 1.0)
    near = (da - db).abs() <= bound
    res["near_ties"] = int(near.sum())
    res["not_near_ties"] = n_bad - res["near_ties"]
    res["largest_gap_over_bound"] = float(((da - db).abs() / bound).max())
    return res
