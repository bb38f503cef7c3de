"""The rule by which the whole-codebook sweeps handle a temperature without dividing every distance, restated in
NumPy and checked against the literal evaluation (vqvae/ema_vector_quantizer.py:49-53: distances / T, argmin).

The kernels compare undivided fp32 distances, keep beside the running minimum the value it displaced last, merge
the partial scans (two lane halves, and the two runs of a sweep that wraps around the end of the codebook) and
call a row "close" when that value lies within 2^-22 |m| of the minimum m (or within T 2^-100, or m / T is not
finite).  A row that is not close must have the literal winner: the first code whose quotient is the smallest.
Close rows are redone literally on the GPU; here they only have to be rare on ordinary data.
"""
import numpy as np
import pytest


def scan(d, order):
    """Sequential scan of the columns `order` (increasing code index) of d [rows, K]: running minimum, its first
    index, and the value it displaced last."""
    rows = d.shape[0]
    best = np.full(rows, np.inf, np.float32)
    sec = np.full(rows, np.inf, np.float32)
    idx = np.zeros(rows, np.int64)
    for k in order:
        v = d[:, k]
        tk = v < best
        sec = np.where(tk, best, sec)
        best = np.where(tk, v, best)
        idx = np.where(tk, k, idx)
    return best, idx, sec


def merge_sec(best_a, sec_a, best_b, sec_b):
    lo, hi = np.minimum(best_a, best_b), np.maximum(best_a, best_b)
    sec = np.minimum(sec_a, sec_b)
    return np.where((hi > lo) & (hi < sec), hi, sec)


def merge(a, b):
    """(best, idx, sec) of two scans over disjoint code sets; ties go to the lower index."""
    take_b = (b[0] < a[0]) | ((b[0] == a[0]) & (b[1] < a[1]))
    return np.where(take_b, b[0], a[0]), np.where(take_b, b[1], a[1]), merge_sec(a[0], a[2], b[0], b[2])


def deferred(d, T, start):
    """The kernels' result for distances d [rows, K] (K a multiple of 64) with the sweep starting at stage `start`."""
    K = d.shape[1]
    stages = K // 64
    halves = []
    for h in (0, 1):
        cols = [k for k in range(K) if (k % 8) // 4 == h]
        hi_run = [k for k in cols if k // 64 >= start]          # swept first: the higher indices
        lo_run = [k for k in cols if k // 64 < start]
        r = scan(d, hi_run)
        if lo_run:
            r = merge(scan(d, lo_run), r)
        halves.append(r)
    m, idx, sec = merge(halves[0], halves[1])
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        q = (m / np.float32(T)).astype(np.float32)
        gap = sec - m
        near = (gap <= np.float32(2.0 ** -22) * np.abs(m)) | (gap <= np.float32(T) * np.float32(2.0 ** -100)) | ~(np.abs(q) < np.inf)
    assert stages >= 1
    return idx, q, near


def literal(d, T):
    with np.errstate(over="ignore", divide="ignore"):
        q = (d / np.float32(T)).astype(np.float32)
    return np.argmin(q, axis=1), q.min(axis=1)


@pytest.mark.parametrize("seed", range(6))
def test_rows_that_are_not_close_have_the_literal_winner(seed):
    rng = np.random.default_rng(9100 + seed)
    rows, K = 20000, 128
    base = rng.uniform(0.5, 600.0, (rows, 1)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 3))
    d = (base * rng.uniform(1.0, 3.0, (rows, K))).astype(np.float32)
    # engineered clusters around the minimum: a few codes within 0 .. 6 units in the last place of each other
    for _ in range(6):
        src = rng.integers(0, K, rows)
        dst = rng.integers(0, K, rows)
        ulps = rng.integers(-3, 4, rows)
        v = d[np.arange(rows), src]
        w = v.copy()
        for _s in range(3):
            w = np.where(ulps > 0, np.nextafter(w, np.float32(np.inf)), np.where(ulps < 0, np.nextafter(w, np.float32(-np.inf)), w))
            ulps = ulps - np.sign(ulps)
        d[np.arange(rows), dst] = w
    if seed % 2:
        d[: rows // 10] -= d[: rows // 10].min(axis=1, keepdims=True) * np.float32(1.0000001)      # minima near or below zero
    for T in (2.5, 0.7, 0.013, 60.0, float(np.float32(10.0 ** rng.uniform(-3, 2)))):
        for start in (0, 1):
            idx, q, near = deferred(d, T, start)
            li, lq = literal(d, T)
            ok = ~near
            assert np.array_equal(idx[ok], li[ok]), (T, start, int((idx[ok] != li[ok]).sum()))
            assert np.array_equal(q[ok].view(np.int32), lq[ok].view(np.int32))
    # the engineered clusters make close rows common here; they must not be all of them
    assert near.mean() < 0.9


def test_close_rows_are_rare_on_gaussian_distances():
    rng = np.random.default_rng(77)
    rows, K = 50000, 512
    d = (512.0 + 45.0 * rng.standard_normal((rows, K))).astype(np.float32)
    idx, q, near = deferred(d, 2.5, 0)
    li, lq = literal(d, 2.5)
    assert np.array_equal(idx[~near], li[~near])
    assert near.mean() < 1e-3


def test_overflowing_and_vanishing_quotients_are_close():
    d = np.array([[3.0, 2.0, 5.0] + [9.0] * 61], np.float32)
    _, _, near = deferred(d, 1e-39, 0)          # every quotient overflows: all equal, the literal winner is code 0
    assert near.all() and literal(d, 1e-39)[0][0] == 0
    tiny = np.array([[3e-45, 1e-45, 5.0] + [9.0] * 61], np.float32)
    _, _, near = deferred(tiny, 100.0, 0)       # quotients below the smallest subnormal: both round to zero
    assert near.all() and literal(tiny, 100.0)[0][0] == 0


def test_quotient_index_keys_order_as_the_argmin_does():
    """Rows that several workgroups share fold their winners with an unsigned 64-bit minimum over keys
    (quotient bits made order-preserving) << 32 | index: smaller quotient first, lower index on equal quotients."""
    rng = np.random.default_rng(5)
    q = np.concatenate([rng.standard_normal(2000).astype(np.float32) * np.float32(10.0 ** rng.uniform(-30, 30)),
                        np.array([0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32)])
    q = np.where(q == 0, np.float32(0.0), q)          # the kernels never see -0 (a sum of opposite terms rounds to +0)
    idx = rng.integers(0, 1 << 20, q.size)
    u = q.view(np.uint32).astype(np.uint64)
    u = np.where(u >> np.uint64(31), u ^ np.uint64(0xFFFFFFFF), u ^ np.uint64(0x80000000))
    key = (u << np.uint64(32)) | idx.astype(np.uint64)
    order = np.argsort(key, kind="stable")
    want = np.lexsort((idx, q))                         # by quotient, then by index
    assert np.array_equal(q[order], q[want]) and np.array_equal(idx[order], idx[want])
    # and the quotient comes back bit for bit
    back = key >> np.uint64(32)
    back = np.where(back >> np.uint64(31), back ^ np.uint64(0x80000000), back ^ np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert np.array_equal(back.view(np.float32).view(np.uint32), q.view(np.uint32))
