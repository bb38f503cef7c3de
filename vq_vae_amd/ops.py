"""Host-side operators over libvqhip.so: torch tensors in, HIP kernels on the current stream.

PyTorch is used for device memory and streams only.  Every function here requires CUDA
(ROCm) tensors and raises otherwise; there is no CPU path.  The functions mirror the steps
of the reference's ``EMAVectorQuantizer.forward`` (vqvae/ema_vector_quantizer.py:33-108).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import LAYOUT_NCHW, LAYOUT_ND, check


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the quantizer runs only as HIP kernels on MI355X "
                           f"(got a tensor on '{t.device}')")


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    _require_cuda(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _rows_view(z: torch.Tensor):
    """A channels_last (B, D, H, W) tensor is, in memory, the row-major (N, D) matrix of its latent
    vectors.  Returns that 2-D view (no copy) and the function that gives an (N, D) result z's
    shape and memory format back -- or (None, None) for every other input.  This is what spares a
    channels_last pipeline the permute + contiguous copy of vqvae/ema_vector_quantizer.py:35."""
    if (z.dim() == 4 and z.is_cuda and not z.is_contiguous()
            and z.is_contiguous(memory_format=torch.channels_last)):
        B, C, H, W = z.shape
        rows = z.permute(0, 2, 3, 1).reshape(B * H * W, C)
        if rows.data_ptr() == z.data_ptr():          # a view, as expected
            return rows, (lambda t: t.view(B, H, W, t.shape[-1]).permute(0, 3, 1, 2))
    return None, None


def ema_scalars(K: int, decay: float, epsilon: float):
    """The four fp32 constants of the EMA update, rounded as the reference's Python expressions
    round when they meet an fp32 tensor (vqvae/ema_vector_quantizer.py:71-79)."""
    return (float(np.float32(decay)), float(np.float32(1 - decay)), float(np.float32(epsilon)),
            float(np.float32(K * epsilon)))


@dataclass
class Workspace:
    """Grow-only device buffers reused from call to call (never part of a state_dict).

    ``clean`` says that the last kernel of the previous ``quantize_step`` on this workspace left the control
    words and the next step's histogram bins zeroed, so the step needs no memsets; anything else that
    touches the control block clears it, and the next step then resets explicitly.  ``flip`` selects which
    of the two alternating histogram buffers the next step counts into."""
    device: torch.device
    _bufs: dict = field(default_factory=dict)
    clean: bool = False
    flip: int = 0
    counts_k: int = 0       # codebook size the alternating histogram buffers were zeroed for

    _views: dict = field(default_factory=dict)

    def get(self, name: str, nbytes: int, dtype=torch.uint8) -> torch.Tensor:
        nbytes = max(int(nbytes), 256)
        key = (name, nbytes, dtype)
        view = self._views.get(key)        # the same request as last step: no tensor arithmetic at all
        if view is not None:
            return view
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._bufs[name] = buf
            self._views = {k: v for k, v in self._views.items() if k[0] != name}   # views of the old buffer
        item = torch.empty((), dtype=dtype).element_size()
        view = buf[: nbytes // item * item].view(dtype)
        self._views[key] = view
        return view


@dataclass
class StepResult:
    quantized: torch.Tensor       # straight-through output, layout of z
    loss: torch.Tensor            # 0-dim
    perplexity: torch.Tensor      # 0-dim
    indices: torch.Tensor         # [N] int64
    counts: torch.Tensor          # [K] int32 bit pattern of uint32 counts
    dmin: Optional[torch.Tensor] = None
    #: data-parallel step asked to defer: enqueues the wait for the all-reduce and the kernels behind it
    #: (loss, perplexity, EMA update).  Call it before anything reads those -- work enqueued before the call
    #: runs beside the exchange.
    finish: Optional[object] = None


def _shape_of(z: torch.Tensor, D: int):
    if z.dim() == 4:
        B, C, H, W = z.shape
        if C != D:
            # same exception type the reference's .view(-1, D) raises on a channel mismatch
            raise RuntimeError(f"shape '[-1, {D}]' is invalid for input with {C} channels "
                               f"(expected (B, {D}, H, W))")
        return LAYOUT_NCHW, B * H * W, H * W
    if z.dim() == 2:
        if z.shape[1] != D:
            raise RuntimeError(f"shape '[-1, {D}]' is invalid for input of size {tuple(z.shape)}")
        return LAYOUT_ND, z.shape[0], 1
    raise RuntimeError(f"expected (B, C, H, W) or (N, D) latents, got {tuple(z.shape)}")


def prepare_codebook(E: torch.Tensor, ws: Workspace, reset: bool = True):
    """vq_codebook_prep: squared norms + MFMA image of the codebook.  Returns (image, e2, control).
    ``reset=False`` is for a caller that knows the control words are zero already."""
    lib = _lib.load()
    K, D = E.shape
    image = ws.get("image", lib.vq_image_bytes(K, D), torch.float32)
    e2 = ws.get("e2", lib.vq_norms_bytes(K), torch.float32)
    control = ws.get("control", lib.vq_control_bytes(), torch.int32)
    s = _stream(E.device)
    if reset:
        ws.clean = False
        check(lib.vq_control_reset(_ptr(control), s), "vq_control_reset")
    check(lib.vq_codebook_prep(_ptr(E), K, D, _ptr(image), _ptr(e2), _ptr(control), s), "vq_codebook_prep")
    return image, e2, control


def filter_supported(K: int, D: int, temperature: float, code_base: int, want_quantized: bool) -> bool:
    """Shapes the filtered search covers (the limits of vq_search_filtered)."""
    return (D in (64, 128, 256) and 0 < K <= 8192 and 0.0 < temperature < float("inf")
            and not (want_quantized and code_base != 0))


FILTER_MAX_CODES = 8192   # codes per filtered launch (12 position bits in a key)


def _search_filtered_chunked(z, E, ws: Workspace, temperature: float, code_base: int, want_quantized: bool,
                             want_dmin: bool, events):
    """Filtered search of a codebook with more than FILTER_MAX_CODES rows: one filtered launch per
    range of rows, the ranges' winners merged through the packed (distance, index) keys -- the
    merge of codebook-row shards, on one device."""
    lib = _lib.load()
    K, D = E.shape
    if events is not None:
        events[0].record(torch.cuda.current_stream(z.device))
    best = None
    r = None
    for lo in range(0, K, FILTER_MAX_CODES):
        r = search(z, E[lo: lo + FILTER_MAX_CODES], ws, temperature=temperature, code_base=code_base + lo,
                   want_quantized=False, want_dmin=True, filtered=True)
        packed = pack_dist_index(r["dmin"], r["indices"])
        if best is None:
            best = packed
        else:
            check(lib.vq_packed_min(_ptr(best), _ptr(packed), best.numel(), _stream(z.device)), "vq_packed_min")
    dmin, idx64, idx32 = unpack_dist_index(best, ws, want_dmin=want_dmin)
    q = lp = None
    n_partial = 0
    if want_quantized:
        q, lp, n_partial = gather_st(z, E, idx64, ws)
    if events is not None:
        events[1].record(torch.cuda.current_stream(z.device))
    return {"indices": idx64, "idx32": idx32, "dmin": dmin, "quantized": q, "loss_partial": lp,
            "n_partial": n_partial, "layout": r["layout"], "N": r["N"], "HW": r["HW"], "control": r["control"],
            "zmax_valid": True}


def search(z: torch.Tensor, E: torch.Tensor, ws: Workspace, temperature: float = 1.0, code_base: int = 0,
           want_quantized: bool = True, want_dmin: bool = False, exact: bool = False, prepared=None,
           events=None, allow_split: bool = True, filtered: bool = False):
    """Nearest-codebook search (+ fused gather / straight-through / loss partials).

    ``filtered`` selects vq_search_filtered (bf16 matrix-core filter + exact fp32 verification:
    identical indices, several times faster on large batches) where it is supported, and the
    plain fp32 search elsewhere.

    Returns dict(indices int64 [N], idx32, dmin|None, quantized|None, loss_partial|None, n_partial).
    ``indices``, ``dmin`` and ``quantized`` are fresh tensors; ``idx32`` and ``loss_partial`` are views of
    buffers in ``ws`` that the next search on the same workspace overwrites (they feed the histogram,
    scatter and scalar kernels of the same step) -- clone them to keep them.
    """
    lib = _lib.load()
    if z.dtype != torch.float32:
        z = z.float()
    rows, back = _rows_view(z)
    if rows is not None and z.shape[1] == E.shape[1]:
        r = search(rows, E, ws, temperature=temperature, code_base=code_base, want_quantized=want_quantized,
                   want_dmin=want_dmin, exact=exact, prepared=prepared, events=events, allow_split=allow_split,
                   filtered=filtered)
        if r["quantized"] is not None:
            r["quantized"] = back(r["quantized"])
        return r
    z = _f32c(z, "latents")
    E = _f32c(E, "codebook")
    K, D = E.shape
    layout, N, HW = _shape_of(z, D)
    if (filtered and not exact and K > FILTER_MAX_CODES
            and filter_supported(FILTER_MAX_CODES, D, float(temperature), code_base, want_quantized)):
        return _search_filtered_chunked(z, E, ws, float(temperature), code_base, want_quantized, want_dmin, events)
    image, e2, control = prepared if prepared is not None else prepare_codebook(E, ws)
    dev = z.device
    idx64 = torch.empty(N, dtype=torch.int64, device=dev)
    idx32 = ws.get("idx32", N * 4, torch.int32)
    dmin = torch.empty(N, dtype=torch.float32, device=dev) if want_dmin else None
    q = torch.empty_like(z) if want_quantized else None
    use_filter = (filtered and not exact and filter_supported(K, D, float(temperature), code_base, want_quantized))
    n_partial = (lib.vq_filter_loss_partial_bytes(N) if use_filter else lib.vq_loss_partial_bytes(N)) // 4
    lp = ws.get("loss_partial", n_partial * 4, torch.float32) if want_quantized else None
    s = _stream(dev)
    if exact:
        check(lib.vq_search_exact(_ptr(z), layout, N, D, HW, _ptr(E), _ptr(e2), K, float(temperature), code_base,
                                  _ptr(idx64), _ptr(idx32), _ptr(dmin), _ptr(q), _ptr(lp), s), "vq_search_exact")
    elif use_filter:
        nf_rows = ws.get("nf_rows", N * 4, torch.int32)
        fimage = ws.get("fimage", lib.vq_filter_image_bytes(K, D))
        work = ws.get("fwork", lib.vq_filter_work_bytes(N))
        if events is not None:
            events[0].record(torch.cuda.current_stream(dev))
        check(lib.vq_filter_prep(_ptr(E), _ptr(e2), K, D, _ptr(fimage), s), "vq_filter_prep")
        check(lib.vq_search_filtered(_ptr(z), layout, N, D, HW, _ptr(E), _ptr(image), _ptr(e2), _ptr(fimage), K,
                                     float(temperature),
                                     code_base, _ptr(idx64), _ptr(idx32), _ptr(dmin), _ptr(q), _ptr(lp),
                                     _ptr(nf_rows), _ptr(control), _ptr(work), s), "vq_search_filtered")
        if events is not None:
            events[1].record(torch.cuda.current_stream(dev))
    else:
        nf_rows = ws.get("nf_rows", N * 4, torch.int32)
        split_bytes = lib.vq_search_split_bytes(N, K) if allow_split else 0
        split_ws = ws.get("split", split_bytes) if split_bytes else None
        if events is not None:
            events[0].record(torch.cuda.current_stream(dev))
        check(lib.vq_search(_ptr(z), layout, N, D, HW, _ptr(E), _ptr(image), _ptr(e2), K, float(temperature),
                            code_base, _ptr(idx64), _ptr(idx32), _ptr(dmin), _ptr(q), _ptr(lp), _ptr(nf_rows),
                            _ptr(control), _ptr(split_ws), s), "vq_search")
        if events is not None:
            events[1].record(torch.cuda.current_stream(dev))
    return {"indices": idx64, "idx32": idx32, "dmin": dmin, "quantized": q, "loss_partial": lp,
            "n_partial": n_partial, "layout": layout, "N": N, "HW": HW, "control": control,
            "zmax_valid": not exact}


def histogram(idx: torch.Tensor, K: int, ws: Workspace, out: Optional[torch.Tensor] = None,
              accumulate: bool = False) -> torch.Tensor:
    """counts[k] = #(idx == k) as an int32 tensor holding uint32 bit patterns."""
    lib = _lib.load()
    _require_cuda(idx, "indices")
    if idx.dtype not in (torch.int32, torch.int64):
        idx = idx.long()
    idx = idx.contiguous().view(-1)
    counts = out if out is not None else torch.empty(K, dtype=torch.int32, device=idx.device)
    control = ws.get("control", lib.vq_control_bytes(), torch.int32)
    ws.clean = False      # (the bad-index counter in the control block)
    check(lib.vq_code_histogram(_ptr(idx), 1 if idx.dtype == torch.int64 else 0, idx.numel(), K, _ptr(counts),
                                1 if accumulate else 0, _ptr(control), _stream(idx.device)), "vq_code_histogram")
    return counts


def embed_sums(z: torch.Tensor, idx32: torch.Tensor, K: int, ws: Workspace, code_base: int = 0,
               zmax_valid: bool = False, sorted_rows: bool = True):
    """vq_ema_scatter: returns (slabs [nslab, K, D], nslab).  ``zmax_valid`` says that a search over the
    same ``z`` has just left max|z| in the workspace's control block."""
    lib = _lib.load()
    z = _f32c(z, "latents")
    D = z.shape[1]
    layout, N, HW = _shape_of(z, D)
    if sorted_rows and (layout == LAYOUT_ND or HW == 1) and z.data_ptr() % 16 == 0 and lib.vq_scatter_sorted_supported(K, D):
        # row-major latents: counting sort by code, every row read whole once (one slab)
        dw = ws.get("slabs", K * D * 4, torch.float32)[: K * D].view(1, K, D)
        work = ws.get("sort_work", lib.vq_scatter_sorted_work_bytes(N, K, D))
        control = ws.get("control", lib.vq_control_bytes(), torch.int32)
        check(lib.vq_ema_scatter_sorted(_ptr(z), N, D, _ptr(idx32), K, code_base, 1 if zmax_valid else 0,
                                        _ptr(control), _ptr(dw), _ptr(work), _stream(z.device)), "vq_ema_scatter_sorted")
        return dw, 1
    nslab = lib.vq_scatter_slabs_for(N, K, D, layout, HW)
    slabs = ws.get("slabs", nslab * K * D * 4, torch.float32)[: nslab * K * D].view(nslab, K, D)
    control = ws.get("control", lib.vq_control_bytes(), torch.int32)
    check(lib.vq_ema_scatter(_ptr(z), layout, N, D, HW, _ptr(idx32), K, code_base, 1 if zmax_valid else 0,
                             _ptr(control), _ptr(slabs), nslab, _stream(z.device)), "vq_ema_scatter")
    return slabs, nslab


def code_sums(x: torch.Tensor, idx32: torch.Tensor, K: int, ws: Workspace) -> torch.Tensor:
    """sums[k, :] = sum of the rows of x assigned to code k (the scatter-add of the EMA update, applied
    to any per-latent tensor: (B, C, H, W) in either memory format, or (N, C))."""
    if x.dtype != torch.float32:
        x = x.float()
    rows, _ = _rows_view(x)
    slabs, _ = embed_sums(rows if rows is not None else x, idx32, K, ws)
    return reduce_slabs(slabs)


def reduce_slabs(slabs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    nslab, K, D = slabs.shape
    dw = out if out is not None else torch.empty(K, D, dtype=torch.float32, device=slabs.device)
    check(lib.vq_reduce_slabs(_ptr(slabs), nslab, K, D, _ptr(dw), _stream(slabs.device)), "vq_reduce_slabs")
    return dw


def step_scalars(loss_partial, n_partial: int, N_total: int, D: int, beta: float, counts, K: int, device,
                 want_sq_sum: bool = False):
    """loss = beta * mean((q - z)^2) and perplexity, as 0-dim device tensors."""
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=device) if loss_partial is not None else None
    perp = torch.empty((), dtype=torch.float32, device=device) if counts is not None else None
    sq = torch.empty((), dtype=torch.float64, device=device) if want_sq_sum else None
    check(lib.vq_step_scalars(_ptr(loss_partial), n_partial, N_total, D, float(np.float32(beta)), _ptr(counts), K,
                              _ptr(loss), _ptr(perp), _ptr(sq), _stream(device)), "vq_step_scalars")
    return loss, perp, sq


def ema_update(counts, slabs: torch.Tensor, usage, cluster_size, ema_w, E, decay: float, epsilon: float,
               counts64=None) -> None:
    """In-place EMA update of usage_count, ema_cluster_size, ema_w and the codebook
    (vqvae/ema_vector_quantizer.py:65-89)."""
    lib = _lib.load()
    K, D = E.shape
    d, omd, eps, keps = ema_scalars(K, decay, epsilon)
    s = _stream(E.device)
    if slabs.dim() == 2:
        slabs = slabs.unsqueeze(0)
    check(lib.vq_ema_cluster_update(_ptr(counts), _ptr(counts64), K, d, omd, eps, keps, _ptr(usage),
                                    _ptr(cluster_size), s), "vq_ema_cluster_update")
    check(lib.vq_ema_codebook_update(_ptr(slabs), slabs.shape[0], K, D, d, omd, _ptr(cluster_size), _ptr(ema_w),
                                     _ptr(E), s), "vq_ema_codebook_update")


def backward(grad_q, grad_loss, z, q_st, coef: float) -> torch.Tensor:
    """grad_z = grad_q + grad_loss * coef * (z - q_st), elementwise over the storage: all operands
    are brought to z's memory format first (channels_last stays channels_last)."""
    lib = _lib.load()
    channels_last = (z.dim() == 4 and not z.is_contiguous() and z.is_contiguous(memory_format=torch.channels_last))
    fmt = torch.channels_last if channels_last else torch.contiguous_format

    def same(t, name):
        _require_cuda(t, name)
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous(memory_format=fmt) if t.dim() == 4 else t.contiguous()

    z = same(z, "latents")
    q_st = same(q_st, "quantized")
    grad_z = torch.empty_like(z)
    if z.numel() == 0:
        return grad_z
    gq = None if grad_q is None else same(grad_q, "grad_quantized")
    gl = None if grad_loss is None else _f32c(grad_loss, "grad_loss")
    check(lib.vq_backward(_ptr(gq), _ptr(gl), _ptr(z), _ptr(q_st), z.numel(), float(coef), _ptr(grad_z),
                          _stream(z.device)), "vq_backward")
    return grad_z


def reseed_codes(E, ema_w, cluster_size, usage, targets: torch.Tensor, donors: torch.Tensor, noise: float,
                 scale: float, ws: Workspace, seed: Optional[int] = None) -> None:
    """vq_reseed_codes: targets become a random donor + N(0, noise^2); EMA statistics scaled.
    The seed comes from torch's CPU generator unless given, so torch.manual_seed governs it."""
    lib = _lib.load()
    K, D = E.shape
    if targets.numel() == 0:
        return
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    targets = targets.to(device=E.device, dtype=torch.int64).contiguous()
    donors = donors.to(device=E.device, dtype=torch.int64).contiguous()
    stage = ws.get("reseed", lib.vq_reseed_stage_bytes(targets.numel(), D), torch.float32)
    check(lib.vq_reseed_codes(_ptr(E), _ptr(ema_w), _ptr(cluster_size), _ptr(usage), K, D, _ptr(targets),
                              targets.numel(), _ptr(donors), donors.numel(), float(noise), float(scale), seed,
                              _ptr(stage), _stream(E.device)), "vq_reseed_codes")


def onehot(idx64: torch.Tensor, K: int) -> torch.Tensor:
    """Dense [N, K] fp32 one-hot matrix (vqvae/ema_vector_quantizer.py:54-55)."""
    lib = _lib.load()
    _require_cuda(idx64, "indices")
    idx64 = idx64.contiguous().view(-1)
    N = idx64.numel()
    enc = torch.empty(N, K, dtype=torch.float32, device=idx64.device)
    if N == 0:
        return enc
    check(lib.vq_onehot(_ptr(idx64), N, K, _ptr(enc), _stream(idx64.device)), "vq_onehot")
    return enc


def gather_st(z: torch.Tensor, E: torch.Tensor, idx64: torch.Tensor, ws: Workspace):
    """Standalone gather + straight-through + loss partials.  Returns (q_st, loss_partial, n_partial)."""
    lib = _lib.load()
    if z.dtype != torch.float32:
        z = z.float()
    rows, back = _rows_view(z)
    if rows is not None and z.shape[1] == E.shape[1]:
        q, lp, n_partial = gather_st(rows, E, idx64, ws)
        return back(q), lp, n_partial
    z = _f32c(z, "latents")
    E = _f32c(E, "codebook")
    K, D = E.shape
    layout, N, HW = _shape_of(z, D)
    q = torch.empty_like(z)
    n_partial = lib.vq_search_blocks(N)
    lp = ws.get("loss_partial", lib.vq_loss_partial_bytes(N), torch.float32)
    check(lib.vq_gather_st(_ptr(z), layout, N, D, HW, _ptr(E), K, _ptr(idx64), _ptr(q), _ptr(lp), _stream(z.device)),
          "vq_gather_st")
    return q, lp, n_partial


def gather_project(z: torch.Tensor, E: torch.Tensor, idx64: torch.Tensor, P: torch.Tensor, ws: Workspace):
    """vq_gather_project: out[n] = P[idx[n]] in z's layout (P = the codebook carried through the 1x1
    convolution that follows the quantizer, vqvae/custom_vqgan.py:72) and the loss partials of
    |E[idx[n]] - z[n]|^2.  Returns (out, loss_partial, n_partial)."""
    lib = _lib.load()
    if z.dtype != torch.float32:
        z = z.float()
    rows, back = _rows_view(z)
    if rows is not None and z.shape[1] == E.shape[1]:
        out, lp, n_partial = gather_project(rows, E, idx64, P, ws)
        return back(out), lp, n_partial
    z = _f32c(z, "latents")
    E = _f32c(E, "codebook")
    P = _f32c(P, "projected codebook")
    K, D = E.shape
    if P.dim() != 2 or P.shape[0] != K:
        raise RuntimeError(f"projected codebook must be ({K}, D'), got {tuple(P.shape)}")
    DP = P.shape[1]
    layout, N, HW = _shape_of(z, D)
    shape = (z.shape[0], DP, z.shape[2], z.shape[3]) if layout == LAYOUT_NCHW else (N, DP)
    out = torch.empty(shape, dtype=torch.float32, device=z.device)
    n_partial = lib.vq_search_blocks(N)
    lp = ws.get("loss_partial", lib.vq_loss_partial_bytes(N), torch.float32)
    check(lib.vq_gather_project(_ptr(z), layout, N, D, HW, _ptr(E), K, _ptr(idx64), _ptr(P), DP, _ptr(out), _ptr(lp),
                                _stream(z.device)), "vq_gather_project")
    return out, lp, n_partial


def backward_gather(grad_q, grad_loss, z: torch.Tensor, E: torch.Tensor, idx64: torch.Tensor, coef: float) -> torch.Tensor:
    """vq_backward_gather: grad_z = grad_q + grad_loss * coef * (z - E[idx]), in z's memory format."""
    lib = _lib.load()
    _require_cuda(z, "latents")
    rows, back = _rows_view(z)
    if rows is not None:
        gq = None
        if grad_q is not None:
            gq = grad_q.float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(rows.shape)
        return back(backward_gather(gq, grad_loss, rows, E, idx64, coef))
    z = _f32c(z, "latents")
    E = _f32c(E, "codebook")
    K, D = E.shape
    layout, N, HW = _shape_of(z, D)
    gq = None if grad_q is None else _f32c(grad_q, "grad_quantized")
    if gq is not None and gq.shape != z.shape:
        raise RuntimeError(f"grad_quantized has shape {tuple(gq.shape)}, latents {tuple(z.shape)}")
    gl = None if grad_loss is None else _f32c(grad_loss, "grad_loss")
    grad_z = torch.empty_like(z)
    if z.numel() == 0:
        return grad_z
    check(lib.vq_backward_gather(_ptr(gq), _ptr(gl), _ptr(z), layout, N, D, HW, _ptr(E), K, _ptr(idx64), float(coef),
                                 _ptr(grad_z), _stream(z.device)), "vq_backward_gather")
    return grad_z


def conv1x1_supported(x: torch.Tensor, c_out: int, c_in: int) -> bool:
    """Shapes vq_conv1x1 covers: fp32 GPU rows with 16 / 32 / 64 / 128 / 256 input channels."""
    return x.is_cuda and c_in in (16, 32, 64, 128, 256) and x.dim() in (2, 4) and x.shape[1] == c_in and c_out > 0


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], ws: Workspace,
            image_name: str = "conv_image") -> torch.Tensor:
    """out[n, o] = sum_c W[o, c] x[n, c] + b[o] on the search's MFMA engine (the 1x1 convolutions of
    vqvae/custom_vqgan.py:50-51).  x: (B, C, H, W) in either memory format, or (N, C) rows; the
    result has x's layout.  weight: (O, C) or (O, C, 1, 1)."""
    lib = _lib.load()
    _require_cuda(x, "input")
    if x.dtype != torch.float32:
        x = x.float()
    W = _f32c(weight.detach().reshape(weight.shape[0], -1), "weight")
    c_out, c_in = W.shape
    rows, back = _rows_view(x)
    if rows is not None:
        return back(conv1x1(rows, W, bias, ws, image_name))
    x = _f32c(x, "input")
    if not conv1x1_supported(x, c_out, c_in):
        raise RuntimeError(f"vq_conv1x1 needs 16/32/64/128/256 input channels, got input {tuple(x.shape)} for a "
                           f"({c_out}, {c_in}) weight")
    layout, N, HW = _shape_of(x, c_in)
    b = None if bias is None else _f32c(bias.detach(), "bias")
    image = ws.get(image_name, lib.vq_conv1x1_image_bytes(c_out, c_in), torch.float32)
    s = _stream(x.device)
    check(lib.vq_conv1x1_prep(_ptr(W), _ptr(b), c_out, c_in, _ptr(image), s), "vq_conv1x1_prep")
    shape = (x.shape[0], c_out, x.shape[2], x.shape[3]) if x.dim() == 4 else (N, c_out)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(lib.vq_conv1x1(_ptr(x), layout, N, c_in, HW, _ptr(image), c_out, _ptr(out), s), "vq_conv1x1")
    return out


def conv1x1_wgrad_supported(gy: torch.Tensor, x: torch.Tensor) -> bool:
    """Shapes vq_conv1x1_wgrad covers: fp32 GPU tensors of one shape family -- (B, C, H, W) with H*W a
    multiple of 32 in NCHW memory order, or channels_last / (N, C) rows with channel counts that are
    multiples of 4 -- and at most 1024 channels each."""
    if not (gy.is_cuda and x.is_cuda and gy.dim() == x.dim() and gy.dim() in (2, 4)):
        return False
    if gy.shape[0] != x.shape[0] or gy.shape[2:] != x.shape[2:] or gy.numel() == 0 or x.numel() == 0:
        return False
    c_out, c_in = gy.shape[1], x.shape[1]
    rows_x, _ = _rows_view(x)
    if gy.dim() == 2 or rows_x is not None:
        layout, N, HW = LAYOUT_ND, x.numel() // c_in, 1
    else:
        layout, N, HW = LAYOUT_NCHW, x.numel() // c_in, x.shape[2] * x.shape[3]
    return bool(_lib.load().vq_conv1x1_wgrad_supported(layout, N, c_out, c_in, HW))


def conv1x1_wgrad(gy: torch.Tensor, x: torch.Tensor, ws: Workspace, want_bias: bool = True):
    """Parameter gradients of a 1x1 convolution y = W x + b (vqvae/custom_vqgan.py:50-51 under autograd):
    returns ``(dW, db)`` with dW[o, i] = sum_n gy[n, o] x[n, i] of shape (C_out, C_in) and
    db[o] = sum_n gy[n, o] (None unless ``want_bias``).  gy and x: (B, C, H, W) or (N, C) rows; a
    channels_last x is read in place as rows and gy is brought to the same order."""
    lib = _lib.load()
    _require_cuda(gy, "grad_output")
    _require_cuda(x, "input")
    if not conv1x1_wgrad_supported(gy, x):
        raise RuntimeError(f"vq_conv1x1_wgrad does not take grad_output {tuple(gy.shape)} with input {tuple(x.shape)}")
    c_out, c_in = gy.shape[1], x.shape[1]
    rows_x, _ = _rows_view(x)
    if rows_x is not None:
        x2 = _f32c(rows_x, "input")
        g2 = gy.float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1, c_out)
        layout, N, HW = LAYOUT_ND, x2.shape[0], 1
    else:
        x2 = _f32c(x, "input")
        g2 = _f32c(gy, "grad_output")
        layout, N, HW = _shape_of(x2, c_in)
    work = ws.get("wgrad_work", lib.vq_conv1x1_wgrad_work_bytes(N, c_out, c_in), torch.float32)
    dW = torch.empty((c_out, c_in), dtype=torch.float32, device=x.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=x.device) if want_bias else None
    check(lib.vq_conv1x1_wgrad(_ptr(g2), _ptr(x2), layout, N, c_out, c_in, HW, _ptr(dW), _ptr(db), _ptr(work),
                               _stream(x.device)), "vq_conv1x1_wgrad")
    return dW, db


def quantize_step(z: torch.Tensor, E: torch.Tensor, ema_w, cluster_size, usage, *, beta: float, decay: float,
                  epsilon: float, temperature: float, training: bool, ws: Workspace, want_dmin: bool = False,
                  stats_reduce=None, search_events=None, filtered: bool = False,
                  project: Optional[torch.Tensor] = None, defer_finish: bool = False) -> StepResult:
    """One full forward of the quantizer: search, gather, loss, histogram, perplexity and, in
    training mode, the EMA update of (usage, cluster_size, ema_w, E) in place.

    ``project`` (K x D'), when given, is the codebook carried through the 1x1 convolution that follows
    the quantizer: ``quantized`` in the result is then that convolution's output (gathered rows of
    ``project``) and the N x D tensor of quantized latents is never written.

    ``stats_reduce(counts_i32, dw_or_None, N_local) -> (counts_i64, dw, N_total)`` is the hook of the
    data-parallel mode: it sums the EMA statistics over ranks before the update (the exchange
    sits between :65/:84 and :71-89 of the reference's forward).  The commitment loss stays the
    local-batch mean, as it would under DistributedDataParallel.
    """
    K, D = E.shape
    if z.dtype != torch.float32:
        z = z.float()
    rows, back = _rows_view(z)
    if rows is not None and z.shape[1] == D:
        # channels_last input: the whole step runs on the (N, D) view, the output gets z's format back
        out = quantize_step(rows, E, ema_w, cluster_size, usage, beta=beta, decay=decay, epsilon=epsilon,
                            temperature=temperature, training=training, ws=ws, want_dmin=want_dmin,
                            stats_reduce=stats_reduce, search_events=search_events, filtered=filtered,
                            project=project, defer_finish=defer_finish)
        out.quantized = back(out.quantized)
        return out
    if z.numel() == 0:
        return _empty_step(z, E, ema_w, cluster_size, usage, decay, epsilon, training, ws, stats_reduce, project)
    if stats_reduce is None:
        return _local_step(z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws,
                           want_dmin, search_events, filtered, project)
    if hasattr(stats_reduce, "reduce_packed"):
        return _dp_step(z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws, want_dmin,
                        search_events, filtered, project, stats_reduce, defer_finish)
    r = search(z, E, ws, temperature=temperature, want_quantized=project is None, want_dmin=want_dmin,
               events=search_events, filtered=filtered)
    if project is not None:
        # the codebook is still the one the search saw: the EMA update comes last
        r["quantized"], r["loss_partial"], r["n_partial"] = gather_project(z, E, r["indices"], project, ws)
    N = r["N"]
    counts = ws.get("counts", K * 4, torch.int32)[:K]
    histogram(r["idx32"][:N], K, ws, out=counts)
    loss, perp, _ = step_scalars(r["loss_partial"], r["n_partial"], N, D, beta, None, K, z.device)
    dw = None
    if training:
        slabs, _ = embed_sums(_f32c(z, "latents"), r["idx32"], K, ws, zmax_valid=r["zmax_valid"])
        dw = reduce_slabs(slabs, out=ws.get("dw", K * D * 4, torch.float32)[: K * D].view(K, D))
    counts64, dw, n_total = stats_reduce(counts, dw, N)
    # perplexity and the EMA update come from the reduced statistics, so every rank computes
    # bit-identical values and the replicas stay in step without a broadcast
    _, perp, _ = step_scalars(None, 0, n_total, D, beta, counts64.to(torch.int32), K, z.device)
    if training:
        ema_update(None, dw, usage, cluster_size, ema_w, E, decay, epsilon, counts64=counts64)
    return StepResult(r["quantized"], loss, perp, r["indices"], counts, r["dmin"])


def _local_step(z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws, want_dmin,
                search_events, filtered, project) -> StepResult:
    """The step on one device, in as few launches as it takes: codebook image, search (+ the exact kernel for
    flagged rows), histogram into pre-zeroed bins, EMA sums, one kernel for loss + perplexity + cluster sizes, one
    for the codebook.  The last kernel zeroes the control words and the other histogram buffer for the next
    step, so a step in steady state issues no memsets (for a 4096-vector batch the launches are the cost)."""
    lib = _lib.load()
    K, D = E.shape
    dev = z.device
    s = _stream(dev)
    control = ws.get("control", lib.vq_control_bytes(), torch.int32)
    both = ws.get("counts2", 2 * K * 4, torch.int32)[: 2 * K].view(2, K)
    if not filtered and project is None and search_events is None:
        return _single_call_step(lib, z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws,
                                 want_dmin, control, both, s)
    if not ws.clean or ws.counts_k != K:
        check(lib.vq_control_reset(_ptr(control), s), "vq_control_reset")
        both.zero_()
        ws.counts_k = K
    ws.clean = False                       # until the last kernel of this step has been issued
    counts, counts_next = both[ws.flip], both[1 - ws.flip]
    prepared = prepare_codebook(E, ws, reset=False)
    r = search(z, E, ws, temperature=temperature, want_quantized=project is None, want_dmin=want_dmin,
               events=search_events, filtered=filtered, prepared=prepared)
    if project is not None:
        # the codebook is still the one the search saw: the EMA update comes last
        r["quantized"], r["loss_partial"], r["n_partial"] = gather_project(z, E, r["indices"], project, ws)
    N = r["N"]
    check(lib.vq_code_histogram(_ptr(r["idx32"]), 0, N, K, _ptr(counts), 1, _ptr(control), s), "vq_code_histogram")
    loss = torch.empty((), dtype=torch.float32, device=dev)
    perp = torch.empty((), dtype=torch.float32, device=dev)
    d, omd, eps, keps = ema_scalars(K, decay, epsilon)
    if training:
        slabs, _ = embed_sums(_f32c(z, "latents"), r["idx32"], K, ws, zmax_valid=r["zmax_valid"])
        if slabs.dim() == 2:
            slabs = slabs.unsqueeze(0)
        check(lib.vq_step_finish(_ptr(r["loss_partial"]), r["n_partial"], N, D, float(np.float32(beta)), _ptr(counts), K,
                                 d, omd, eps, keps, _ptr(usage), _ptr(cluster_size), _ptr(loss), _ptr(perp), None, None,
                                 s), "vq_step_finish")
        check(lib.vq_ema_codebook_update_end(_ptr(slabs), slabs.shape[0], K, D, d, omd, _ptr(cluster_size), _ptr(ema_w),
                                             _ptr(E), _ptr(control), _ptr(counts_next), s), "vq_ema_codebook_update_end")
    else:
        check(lib.vq_step_finish(_ptr(r["loss_partial"]), r["n_partial"], N, D, float(np.float32(beta)), _ptr(counts), K,
                                 d, omd, eps, keps, None, None, _ptr(loss), _ptr(perp), _ptr(control), _ptr(counts_next),
                                 s), "vq_step_finish")
    ws.clean = True
    ws.flip = 1 - ws.flip
    return StepResult(r["quantized"], loss, perp, r["indices"], counts, r["dmin"])


def _dp_step(z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws, want_dmin, search_events,
             filtered, project, stats_reduce, defer_finish) -> StepResult:
    """The data-parallel step: search and local statistics, vq_pack_stats, ONE fp32 all-reduce, then
    vq_step_finish_reduced and the codebook update from the reduced buffer -- every rank applies the same
    update, so the replicas stay bit-identical without a broadcast.  Packing, unpacking, the row total and the
    housekeeping are the library's kernels; the host only issues the collective."""
    lib = _lib.load()
    K, D = E.shape
    dev = z.device
    s = _stream(dev)
    control = ws.get("control", lib.vq_control_bytes(), torch.int32)
    both = ws.get("counts2", 2 * K * 4, torch.int32)[: 2 * K].view(2, K)
    if not ws.clean or ws.counts_k != K:
        check(lib.vq_control_reset(_ptr(control), s), "vq_control_reset")
        both.zero_()
        ws.counts_k = K
    ws.clean = False
    counts, counts_next = both[ws.flip], both[1 - ws.flip]
    prepared = prepare_codebook(E, ws, reset=False)
    r = search(z, E, ws, temperature=temperature, want_quantized=project is None, want_dmin=want_dmin,
               events=search_events, filtered=filtered, prepared=prepared)
    if project is not None:
        r["quantized"], r["loss_partial"], r["n_partial"] = gather_project(z, E, r["indices"], project, ws)
    N = r["N"]
    check(lib.vq_code_histogram(_ptr(r["idx32"]), 0, N, K, _ptr(counts), 1, _ptr(control), s), "vq_code_histogram")
    slabs, nslab = None, 0
    if training:
        slabs, nslab = embed_sums(_f32c(z, "latents"), r["idx32"], K, ws, zmax_valid=r["zmax_valid"])
    kd = K * D if training else 0
    buf = ws.get("dp_buf", lib.vq_packed_stats_floats(K, D, 1 if training else 0) * 4, torch.float32)
    check(lib.vq_pack_stats(_ptr(slabs), nslab, K, D, _ptr(counts), N, _ptr(buf), s), "vq_pack_stats")
    wait = stats_reduce.reduce_packed(buf)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    perp = torch.empty((), dtype=torch.float32, device=dev)
    d, omd, eps, keps = ema_scalars(K, decay, epsilon)
    lp, n_partial = r["loss_partial"], r["n_partial"]

    def finish():
        wait()
        tail = ctypes.c_void_p(buf.data_ptr() + kd * 4)
        if training:
            check(lib.vq_step_finish_reduced(_ptr(lp), n_partial, N, D, float(np.float32(beta)), tail, K, d, omd, eps,
                                             keps, _ptr(usage), _ptr(cluster_size), _ptr(loss), _ptr(perp), None, None,
                                             None, None, s), "vq_step_finish_reduced")
            check(lib.vq_ema_codebook_update_end(_ptr(buf), 1, K, D, d, omd, _ptr(cluster_size), _ptr(ema_w), _ptr(E),
                                                 _ptr(control), _ptr(counts_next), s), "vq_ema_codebook_update_end")
        else:
            check(lib.vq_step_finish_reduced(_ptr(lp), n_partial, N, D, float(np.float32(beta)), tail, K, d, omd, eps,
                                             keps, None, None, _ptr(loss), _ptr(perp), None, None, _ptr(control),
                                             _ptr(counts_next), s), "vq_step_finish_reduced")
        ws.clean = True
        ws.flip = 1 - ws.flip

    out = StepResult(r["quantized"], loss, perp, r["indices"], counts, r["dmin"])
    if defer_finish:
        out.finish = finish
    else:
        finish()
    return out


def _single_call_step(lib, z, E, ema_w, cluster_size, usage, beta, decay, epsilon, temperature, training, ws, want_dmin,
                      control, both, s) -> StepResult:
    """The same step through vq_quantize_step: one call into the library, which enqueues the kernels back to
    back (the Python side of a call costs as much as a small kernel runs)."""
    z = _f32c(z, "latents")
    E = _f32c(E, "codebook")
    K, D = E.shape
    dev = z.device
    layout, N, HW = _shape_of(z, D)
    a = ws._views.get("step_args")
    if a is None:
        a = ws._views["step_args"] = _lib.StepArgs()
    idx64 = torch.empty(N, dtype=torch.int64, device=dev)
    q = torch.empty_like(z)
    scal = torch.empty(2, dtype=torch.float32, device=dev)
    dmin = torch.empty(N, dtype=torch.float32, device=dev) if want_dmin else None
    idx32 = ws.get("idx32", N * 4, torch.int32)
    split_bytes = lib.vq_search_split_bytes(N, K)
    d, omd, eps, keps = ema_scalars(K, decay, epsilon)
    a.z, a.layout, a.N, a.D, a.HW = z.data_ptr(), layout, N, D, HW
    a.E, a.K = E.data_ptr(), K
    a.beta, a.decay, a.one_minus_decay, a.eps, a.k_eps = float(np.float32(beta)), d, omd, eps, keps
    a.temperature, a.training = float(temperature), 1 if training else 0
    a.idx64, a.q_out, a.loss_out, a.perplexity_out = idx64.data_ptr(), q.data_ptr(), scal.data_ptr(), scal.data_ptr() + 4
    a.dmin = dmin.data_ptr() if dmin is not None else None
    a.image = ws.get("image", lib.vq_image_bytes(K, D), torch.float32).data_ptr()
    a.e2 = ws.get("e2", lib.vq_norms_bytes(K), torch.float32).data_ptr()
    a.control, a.idx32 = control.data_ptr(), idx32.data_ptr()
    a.nf_rows = ws.get("nf_rows", N * 4, torch.int32).data_ptr()
    a.loss_partial = ws.get("loss_partial", lib.vq_loss_partial_bytes(N), torch.float32).data_ptr()
    a.split_ws = ws.get("split", split_bytes).data_ptr() if split_bytes else None
    clean = ws.clean and ws.counts_k == K      # otherwise the library zeroes control and this step's bins first
    ws.counts_k = K
    counts, counts_next = both[ws.flip], both[1 - ws.flip]
    a.counts, a.counts_next = counts.data_ptr(), counts_next.data_ptr()
    a.sort_work, a.slabs, a.nslab = None, None, 0
    if training:
        a.ema_w, a.cluster_size, a.usage = ema_w.data_ptr(), cluster_size.data_ptr(), usage.data_ptr()
        if (layout == LAYOUT_ND or HW == 1) and z.data_ptr() % 16 == 0 and lib.vq_scatter_sorted_supported(K, D):
            a.nslab = 1
            a.sort_work = ws.get("sort_work", lib.vq_scatter_sorted_work_bytes(N, K, D)).data_ptr()
        else:
            a.nslab = lib.vq_scatter_slabs_for(N, K, D, layout, HW)
        a.slabs = ws.get("slabs", a.nslab * K * D * 4, torch.float32).data_ptr()
    a.control_clean = 1 if clean else 0
    ws.clean = False
    check(lib.vq_quantize_step(ctypes.byref(a), s), "vq_quantize_step")
    ws.clean = True
    ws.flip = 1 - ws.flip
    return StepResult(q, scal[0], scal[1], idx64, counts, dmin)


def _empty_step(z, E, ema_w, cluster_size, usage, decay, epsilon, training, ws, stats_reduce, project) -> StepResult:
    """A batch without rows, as the reference's forward treats it: empty outputs, loss and perplexity
    NaN (means over nothing, vqvae/ema_vector_quantizer.py:92, :105), and in training mode the EMA
    update with all-zero statistics (:67-89: the cluster sizes and sums decay, a fresh codebook
    becomes 0/0)."""
    K, D = E.shape
    dev = z.device
    shape = list(z.shape)
    if project is not None:
        shape[1] = project.shape[1]
    q = torch.empty(shape, dtype=torch.float32, device=dev)
    nan = torch.full((), float("nan"), dtype=torch.float32, device=dev)
    counts = ws.get("counts", K * 4, torch.int32)[:K]
    counts.zero_()
    if stats_reduce is not None and hasattr(stats_reduce, "reduce_packed"):
        # this rank has no rows but the others do: take part in the same exchange with all-zero statistics
        lib = _lib.load()
        s = _stream(dev)
        kd = K * D if training else 0
        buf = ws.get("dp_buf", lib.vq_packed_stats_floats(K, D, 1 if training else 0) * 4, torch.float32)
        buf.zero_()
        stats_reduce.reduce_packed(buf)()
        perp = torch.empty((), dtype=torch.float32, device=dev)
        d, omd, eps, keps = ema_scalars(K, decay, epsilon)
        tail = ctypes.c_void_p(buf.data_ptr() + kd * 4)
        check(lib.vq_step_finish_reduced(None, 0, 0, D, 0.0, tail, K, d, omd, eps, keps, _ptr(usage) if training else None,
                                         _ptr(cluster_size) if training else None, None, _ptr(perp), None, None, None, None,
                                         s), "vq_step_finish_reduced")
        if training:
            check(lib.vq_ema_codebook_update_end(_ptr(buf), 1, K, D, d, omd, _ptr(cluster_size), _ptr(ema_w), _ptr(E),
                                                 None, None, s), "vq_ema_codebook_update_end")
        ws.clean = False
        return StepResult(q, nan, perp, torch.empty(0, dtype=torch.int64, device=dev), counts, None)
    if training:
        dw = ws.get("dw", K * D * 4, torch.float32)[: K * D].view(K, D)
        dw.zero_()
        if stats_reduce is None:
            ema_update(counts, dw, usage, cluster_size, ema_w, E, decay, epsilon)
        else:
            counts64, dw, _ = stats_reduce(counts, dw, 0)
            ema_update(None, dw, usage, cluster_size, ema_w, E, decay, epsilon, counts64=counts64)
    elif stats_reduce is not None:
        stats_reduce(counts, None, 0)       # keep the collective sequence of the other ranks
    return StepResult(q, nan, nan.clone(), torch.empty(0, dtype=torch.int64, device=dev), counts, None)


def pack_dist_index(dmin: torch.Tensor, idx64: torch.Tensor) -> torch.Tensor:
    """int64 keys whose signed order is torch.argmin's order over (distance, index)."""
    lib = _lib.load()
    N = idx64.numel()
    packed = torch.empty(N, dtype=torch.int64, device=idx64.device)
    check(lib.vq_pack_dist_index(_ptr(dmin), _ptr(idx64), N, _ptr(packed), _stream(idx64.device)),
          "vq_pack_dist_index")
    return packed


def unpack_dist_index(packed: torch.Tensor, ws: Workspace, want_dmin: bool = True):
    """Returns (dmin|None, idx64, idx32) of the winners in ``packed``."""
    lib = _lib.load()
    N = packed.numel()
    dev = packed.device
    dmin = torch.empty(N, dtype=torch.float32, device=dev) if want_dmin else None
    idx64 = torch.empty(N, dtype=torch.int64, device=dev)
    idx32 = ws.get("idx32", N * 4, torch.int32)
    check(lib.vq_unpack_dist_index(_ptr(packed), N, _ptr(dmin), _ptr(idx64), _ptr(idx32), _stream(dev)),
          "vq_unpack_dist_index")
    return dmin, idx64, idx32
