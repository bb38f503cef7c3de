"""ctypes front end of the CPU oracle (oracle/vq_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests, the smoke check and the benchmark's cpu_baseline leg import this module;
the product package never does.  Every function works on NumPy arrays and follows the
reference lines cited in vq_oracle.c.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvqoracle.so")

ND, NCHW = 0, 1


def build(force: bool = False) -> str:
    """Compile oracle/vq_oracle.c with gcc (make) when the library is missing or stale."""
    src = os.path.join(_HERE, "vq_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libvqoracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.vqo_perplexity.restype = ctypes.c_float
        _lib.vqo_version.restype = ctypes.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def codebook_norms(E: np.ndarray) -> np.ndarray:
    E = _f32(E)
    K, D = E.shape
    e2 = np.empty(K, np.float32)
    lib().vqo_codebook_norms(_p(E), ctypes.c_int(K), ctypes.c_int(D), _p(e2))
    return e2


def search(z: np.ndarray, E: np.ndarray, T: float = 1.0, code_base: int = 0, skip_self: bool = False,
           e2: np.ndarray | None = None):
    """z: (B, D, H, W) NCHW or (N, D).  Returns (idx int64 [N], dmin float32 [N])."""
    z = _f32(z)
    E = _f32(E)
    K, D = E.shape
    if z.ndim == 4:
        B, C, H, W = z.shape
        assert C == D
        layout, N, HW = NCHW, B * H * W, H * W
    else:
        assert z.ndim == 2 and z.shape[1] == D
        layout, N, HW = ND, z.shape[0], 1
    if e2 is None:
        e2 = codebook_norms(E)
    e2 = _f32(e2)
    idx = np.empty(N, np.int64)
    dmin = np.empty(N, np.float32)
    lib().vqo_search(_p(z), ctypes.c_int(layout), ctypes.c_int64(N), ctypes.c_int(D), ctypes.c_int64(HW),
                     _p(E), _p(e2), ctypes.c_int(K), ctypes.c_float(T), ctypes.c_int64(code_base),
                     ctypes.c_int(1 if skip_self else 0), _p(idx), _p(dmin))
    return idx, dmin


def gather_st(z: np.ndarray, E: np.ndarray, idx: np.ndarray):
    """Returns (q_st with z's layout, sum of squared differences as float64)."""
    z = _f32(z)
    E = _f32(E)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    K, D = E.shape
    if z.ndim == 4:
        B, C, H, W = z.shape
        layout, N, HW = NCHW, B * H * W, H * W
    else:
        layout, N, HW = ND, z.shape[0], 1
    q = np.empty_like(z)
    sq = ctypes.c_double(0.0)
    lib().vqo_gather_st(_p(z), ctypes.c_int(layout), ctypes.c_int64(N), ctypes.c_int(D), ctypes.c_int64(HW),
                        _p(E), _p(idx), _p(q), ctypes.byref(sq))
    return q, sq.value


def histogram(idx: np.ndarray, K: int) -> np.ndarray:
    idx = np.ascontiguousarray(idx, dtype=np.int64).ravel()
    counts = np.empty(K, np.int64)
    lib().vqo_histogram(_p(idx), ctypes.c_int64(idx.size), ctypes.c_int(K), _p(counts))
    return counts


def embed_sum(z: np.ndarray, idx: np.ndarray, K: int) -> np.ndarray:
    z = _f32(z)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if z.ndim == 4:
        B, D, H, W = z.shape
        layout, N, HW = NCHW, B * H * W, H * W
    else:
        layout, N, HW = ND, z.shape[0], 1
        D = z.shape[1]
    dw = np.empty((K, D), np.float32)
    lib().vqo_embed_sum(_p(z), ctypes.c_int(layout), ctypes.c_int64(N), ctypes.c_int(D), ctypes.c_int64(HW),
                        _p(idx), ctypes.c_int(K), _p(dw))
    return dw


def perplexity(counts: np.ndarray, N: int) -> np.float32:
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    return np.float32(lib().vqo_perplexity(_p(counts), ctypes.c_int(counts.size), ctypes.c_int64(N)))


def ema_scalars(K: int, decay: float, epsilon: float):
    """The four fp32 constants of the EMA update, rounded the way the reference's Python
    expressions round them (ema_vector_quantizer.py:71-79)."""
    return (np.float32(decay), np.float32(1 - decay), np.float32(epsilon), np.float32(K * epsilon))


def ema_update(counts, dw, usage, cs, ema_w, E, decay: float, epsilon: float):
    """In-place EMA update of usage, cs, ema_w, E (all float32 arrays)."""
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    dw = _f32(dw)
    K, D = E.shape
    for a in (usage, cs, ema_w, E):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    d, omd, eps, keps = ema_scalars(K, decay, epsilon)
    lib().vqo_ema_update(_p(counts), _p(dw), ctypes.c_int(K), ctypes.c_int(D), ctypes.c_float(d),
                         ctypes.c_float(omd), ctypes.c_float(eps), ctypes.c_float(keps), _p(usage), _p(cs),
                         _p(ema_w), _p(E))


def cluster_stats(cs: np.ndarray, usage: np.ndarray) -> dict:
    cs = _f32(cs)
    usage = _f32(usage)
    out = np.empty(5, np.float64)
    lib().vqo_cluster_stats(_p(cs), _p(usage), ctypes.c_int(cs.size), _p(out))
    return {"active_size": int(out[0]), "utilization": float(out[1]), "entropy": float(out[2]),
            "normalized_entropy": float(out[3]), "total_usage": float(out[4])}


def index_stats_f64(counts: np.ndarray, M: int) -> dict:
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty(5, np.float64)
    lib().vqo_index_stats_f64(_p(counts), ctypes.c_int(counts.size), ctypes.c_int64(M), _p(out))
    return {"unique": int(out[0]), "utilization": float(out[1]), "entropy": float(out[2]),
            "normalized_entropy": float(out[3]), "gini": float(out[4])}


def index_stats_f32(counts: np.ndarray) -> dict:
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty(4, np.float32)
    lib().vqo_index_stats_f32(_p(counts), ctypes.c_int(counts.size), _p(out))
    return {"unique": int(out[0]), "utilization": float(out[1]), "entropy": float(out[2]),
            "normalized_entropy": float(out[3])}


def pairwise(E: np.ndarray):
    E = _f32(E)
    K, D = E.shape
    nn_dist = np.empty(K, np.float32)
    nn_idx = np.empty(K, np.int64)
    out = np.empty(3, np.float64)
    lib().vqo_pairwise(_p(E), ctypes.c_int(K), ctypes.c_int(D), _p(nn_dist), _p(nn_idx), _p(out))
    return nn_dist, nn_idx, {"min": float(out[0]), "mean": float(out[1]), "max": float(out[2])}


@dataclass
class OracleState:
    """The six state tensors of the reference module (ema_vector_quantizer.py:21-31)."""
    embedding: np.ndarray            # [K, D]  embedding.weight
    ema_cluster_size: np.ndarray     # [K]
    ema_w: np.ndarray                # [K, D]
    usage_count: np.ndarray          # [K]

    @staticmethod
    def fresh(E: np.ndarray) -> "OracleState":
        E = _f32(E).copy()
        K, D = E.shape
        return OracleState(E, np.zeros(K, np.float32), np.zeros((K, D), np.float32), np.zeros(K, np.float32))

    def copy(self) -> "OracleState":
        return OracleState(self.embedding.copy(), self.ema_cluster_size.copy(), self.ema_w.copy(),
                           self.usage_count.copy())


def forward(z: np.ndarray, st: OracleState, commitment_cost: float, decay: float = 0.95, epsilon: float = 1e-5,
            temperature: float = 1.0, training: bool = True):
    """One EMAVectorQuantizer.forward (ema_vector_quantizer.py:33-108) on an NCHW batch.

    Returns dict(quantized, loss, perplexity, indices [N,1], counts) and mutates ``st`` in
    training mode.  The gather uses the pre-update codebook (:58 precedes :89).
    """
    z = _f32(z)
    B, D, H, W = z.shape
    N = B * H * W
    K = st.embedding.shape[0]
    idx, dmin = search(z, st.embedding, T=temperature)
    q_st, sq = gather_st(z, st.embedding, idx)
    counts = histogram(idx, K)
    if training:
        dw = embed_sum(z, idx, K)
        ema_update(counts, dw, st.usage_count, st.ema_cluster_size, st.ema_w, st.embedding, decay, epsilon)
    loss = np.float32(np.float32(commitment_cost) * np.float32(sq / (N * D)))
    return {"quantized": q_st, "loss": loss, "perplexity": perplexity(counts, N), "indices": idx.reshape(N, 1),
            "counts": counts, "dmin": dmin}


def conv1x1(x: np.ndarray, W: np.ndarray, b) -> np.ndarray:
    """nn.Conv2d(C, O, kernel_size=1) as the reference places it either side of the quantizer
    (custom_vqgan.py:50-51): out[n, o, h, w] = sum_i W[o, i] x[n, i, h, w] + b[o]; float64
    accumulation, float32 result."""
    W2 = np.asarray(W, np.float64).reshape(W.shape[0], -1)
    out = np.einsum("oi,nihw->nohw", W2, np.asarray(x, np.float64))
    if b is not None:
        out = out + np.asarray(b, np.float64)[None, :, None, None]
    return out.astype(np.float32)


def boundary(h: np.ndarray, Wq, bq, Wp, bp, st: OracleState, commitment_cost: float, decay: float = 0.95,
             epsilon: float = 1e-5, temperature: float = 1.0, training: bool = True, grad_out=None,
             grad_loss=None):
    """quant_conv -> quantizer -> post_quant_conv (custom_vqgan.py:68, :81, :72) and, when grad_out /
    grad_loss are given, the gradients autograd produces for them: the straight-through output passes
    its gradient to the latents unchanged (ema_vector_quantizer.py:95), the commitment loss adds
    grad_loss * 2 beta / (N D) * (z - q) (:92), the codebook receives nothing."""
    h = _f32(h)
    z = conv1x1(h, Wq, bq)
    r = forward(z, st, commitment_cost, decay, epsilon, temperature, training)
    q = r["quantized"]
    res = {"out": conv1x1(q, Wp, bp), "loss": r["loss"], "perplexity": r["perplexity"], "indices": r["indices"],
           "z": z}
    if grad_out is None and grad_loss is None:
        return res
    B, D, H, W_ = z.shape
    go = np.zeros(res["out"].shape, np.float64) if grad_out is None else np.asarray(grad_out, np.float64)
    gl = 0.0 if grad_loss is None else float(grad_loss)
    Wp2 = np.asarray(Wp, np.float64).reshape(Wp.shape[0], -1)
    Wq2 = np.asarray(Wq, np.float64).reshape(Wq.shape[0], -1)
    q64, z64, h64 = q.astype(np.float64), z.astype(np.float64), h.astype(np.float64)
    gz = np.einsum("oi,nohw->nihw", Wp2, go) + gl * 2.0 * float(commitment_cost) / z.size * (z64 - q64)
    res.update(grad_Wp=np.einsum("nohw,nihw->oi", go, q64).reshape(np.shape(Wp)).astype(np.float32),
               grad_bp=go.sum((0, 2, 3)).astype(np.float32),
               grad_z=gz.astype(np.float32),
               grad_h=np.einsum("oi,nohw->nihw", Wq2, gz).astype(np.float32),
               grad_Wq=np.einsum("nohw,nihw->oi", gz, h64).reshape(np.shape(Wq)).astype(np.float32),
               grad_bq=gz.sum((0, 2, 3)).astype(np.float32))
    return res
