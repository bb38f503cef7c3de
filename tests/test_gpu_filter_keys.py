"""Direct checks of what the filtered search assumes about the bf16 matrix core.

1. The accumulation error of the instruction chain itself (tests/hip/mfma_probe.hip runs the filter
   kernel's chain of v_mfma_f32_32x32x16_bf16 on given operands): measured against the exact sum of
   products, it must stay well inside the allowance the error bound grants it.
2. The keys the filter kernel leaves in its work buffer, decoded together with the bf16 codebook
   image and the per-row terms: every key is a lower bound of the shifted true squared distance, and
   differs from the exact value of its own expression by no more than the allowance plus the key
   truncation.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from inputs import gauss

pytestmark = pytest.mark.gpu
f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def bf16_rn_bits(x):
    u = np.asarray(x, f32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def bf16_to_f32(bits):
    return (np.asarray(bits, np.uint16).astype(np.uint32) << 16).view(f32)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


PROBE_CASES = {
    "gauss": lambda kk: (gauss(1, (32, kk)), gauss(2, (32, kk))),
    "all positive": lambda kk: (np.abs(gauss(3, (32, kk))), np.abs(gauss(4, (32, kk)))),
    "wide range": lambda kk: (gauss(5, (32, kk)) * np.exp(3 * gauss(6, (32, kk))), gauss(7, (32, kk))),
    "cancelling": lambda kk: (np.tile(np.array([1.0, -1.0], f32), (32, kk // 2)) * (1 + f32(1e-2) * gauss(8, (32, kk))),
                              np.ones((32, kk), f32) * f32(3.0)),
    "decaying": lambda kk: (gauss(9, (32, kk)) * (f32(0.9) ** np.arange(kk, dtype=f32)), gauss(10, (32, kk))),
    "growing": lambda kk: (gauss(11, (32, kk)) * (f32(1.1) ** np.arange(kk, dtype=f32)), gauss(12, (32, kk))),
}


@pytest.mark.parametrize("name", list(PROBE_CASES))
@pytest.mark.parametrize("kk", [80, 272])
def test_matrix_core_accumulation_error(name, kk, dev):
    """|MFMA chain - exact| <= 2 * kk * 2^-24 * sum_k |a_k b_k|: twice the textbook bound of a
    round-to-nearest fp32 summation, a quarter of the 7-ulp-per-product allowance of the filter's bound
    ((D + 24) 2^-21 with kk = D + 16)."""
    path = os.path.join(HERE, "hip", "libmfma_probe.so")
    if not os.path.exists(path):      # normally built beforehand by __graft_entry__.build()
        import sys
        sys.path.insert(0, os.path.join(HERE, "hip"))
        import build_probe
        build_probe.build()
    lib = ctypes.CDLL(path)
    lib.mfma_probe_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    a, b = PROBE_CASES[name](kk)
    ab, bb = bf16_rn_bits(a.astype(f32)), bf16_rn_bits(b.astype(f32))
    A = torch.from_numpy(ab.view(np.int16)).to(dev)
    B = torch.from_numpy(bb.view(np.int16)).to(dev)
    C = torch.empty(32, 32, dtype=torch.float32, device=dev)
    rc = lib.mfma_probe_bf16(A.data_ptr(), B.data_ptr(), kk, C.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    a64, b64 = bf16_to_f32(ab).astype(np.float64), bf16_to_f32(bb).astype(np.float64)
    exact = a64 @ b64.T
    mag = np.abs(a64) @ np.abs(b64).T
    err = np.abs(C.cpu().numpy().astype(np.float64) - exact)
    ratio = err / (kk * 2.0 ** -24 * mag)
    print(f"{name} kk={kk}: worst error = {ratio.max():.3f} x (kk 2^-24 sum|ab|)")
    assert ratio.max() <= 2.0


KEY_CASES = {
    "gauss D=256": lambda: (gauss(21, (1024, 256)), gauss(22, (512, 256))),
    "gauss D=64": lambda: (gauss(23, (1024, 64)), gauss(24, (384, 64)) * f32(0.7)),
    "offset": lambda: (gauss(25, (1024, 128)) + f32(5), gauss(26, (512, 128)) + f32(5)),
    "wide range": lambda: (gauss(27, (1024, 64)) * np.exp(gauss(28, (1024, 1))), gauss(29, (256, 64)) * np.exp(gauss(30, (256, 1)))),
}


@pytest.mark.parametrize("name", list(KEY_CASES))
def test_filter_keys_are_lower_bounds(name, dev):
    from vq_vae_amd import ops
    z, E = KEY_CASES[name]()
    z, E = np.ascontiguousarray(z, f32), np.ascontiguousarray(E, f32)
    N, D = z.shape
    K = E.shape[0]
    ws = ops.Workspace(dev)
    zt_, Et_ = torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev)
    r = ops.search(zt_, Et_, ws, want_quantized=False, filtered=True)
    torch.cuda.synchronize()
    # ---- decode the filter's buffers (layouts: vq_vae_amd/csrc/vq_filter.hip)
    FS = D // 16 + 1
    tiles = (K + 127) // 128 * 4
    fim = ws._bufs["fimage"].cpu().numpy()
    img = fim[: tiles * FS * 1024].view(np.uint16).reshape(tiles, FS, 64, 8)
    vec = np.zeros((tiles * 32, D), f32)                       # -2 * e~' per code
    for s in range(D // 16):
        for g in range(2):
            vec[:, 16 * s + 8 * g: 16 * s + 8 * g + 8] = bf16_to_f32(img[:, s, 32 * g: 32 * g + 32, :]).reshape(tiles * 32, 8)
    last0 = bf16_to_f32(img[:, FS - 1, :32, :]).reshape(tiles * 32, 8).astype(np.float64)
    last1 = bf16_to_f32(img[:, FS - 1, 32:, :]).reshape(tiles * 32, 8).astype(np.float64)
    n2 = last0[:, 0] + last0[:, 1] + last0[:, 2]
    assert np.all(last0[:K, 3] == 1.0)
    pde, pen, pe2, pun, pu2 = last0[:, 4], last0[:, 5], last0[:, 6], last0[:, 7], last1[:, 0]
    mu_off = tiles * FS * 1024 + 256 + tiles * 32 * 32
    mu = fim[mu_off: mu_off + 4 * D].view(f32)
    work = ws._bufs["fwork"].cpu().numpy()
    NP = (N + 3) // 4 * 4
    keys = work[: NP * 32].view(np.uint32).reshape(NP, 2, 4)[:N]
    raux = work[NP * 32: NP * 64].view(f32).reshape(NP, 2, 4)[:N].astype(np.float64)
    cst, c1, c2, c3, c4, c5 = raux[:, 0, 0], raux[:, 0, 1], raux[:, 0, 2], raux[:, 0, 3], raux[:, 1, 0], raux[:, 1, 1]
    # ---- the row side as the kernel forms it
    z1 = (z - mu).astype(f32)
    zb = bf16_to_f32(bf16_rn_bits(z1)).astype(np.float64)
    g_of = np.array([0, 1])[None, :, None]
    reg = keys & 15
    code = ((keys >> 4) & 255) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * g_of
    val = (keys & np.uint32(0xFFFFF000)).view(f32).astype(np.float64)
    ok = np.isfinite(val) & (code < K)
    assert ok[:, :, 0].all(), "every lane half has seen at least one code"
    cc = np.where(ok, code, 0)
    rest = (n2[cc] + np.einsum("nd,ngrd->ngr", zb, vec[cc].astype(np.float64))
            - (c1[:, None, None] * pde[cc] + c2[:, None, None] * pen[cc] + c3[:, None, None] * pe2[cc]
               + c4[:, None, None] * pun[cc] + c5[:, None, None] * pu2[cc]))
    # the per-row shift is one bf16 number: the keys of a row agree on it once snapped to the bf16 grid
    d = np.where(ok, val - rest, np.nan)
    cup = bf16_to_f32(bf16_rn_bits(np.nanmedian(d.reshape(N, 8), axis=1).astype(f32))).astype(np.float64)
    exact = rest + cup[:, None, None]
    e_norm = 0.5 * np.sqrt((vec.astype(np.float64) ** 2).sum(1))
    z_norm = np.sqrt((zb ** 2).sum(1))
    allow = (D + 24) * 2.0 ** -21 * (z_norm[:, None, None] + e_norm[cc]) ** 2
    over = np.where(ok, val - exact, -np.inf)
    under = np.where(ok, exact * (1 - 2.0 ** -11) - val, -np.inf)
    print(f"{name}: key - exact <= {np.max(over / allow):.3f} x allowance, truncated by <= {np.max(under / allow):.3f} x allowance")
    assert (over <= allow).all()
    assert (under <= allow).all()
    # ---- and against the true distances: key <= ||z - e||^2 + (shift - ||z'||^2), with z' = z - mu exactly
    z64, E64, mu64 = z.astype(np.float64), E.astype(np.float64), mu.astype(np.float64)
    dist = ((z64[:, None, None, :] - E64[cc]) ** 2).sum(-1)
    shift = cup - ((z64 - mu64) ** 2).sum(1)
    assert np.where(ok, val - (dist + shift[:, None, None]), -np.inf).max() <= 0.0
    # the winners themselves equal the fp32 search's
    ref = ops.search(zt_, Et_, ws, want_quantized=False)["indices"]
    assert torch.equal(r["indices"], ref)
