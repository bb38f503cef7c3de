"""The error bound of the filtered search, restated in NumPy and checked on the CPU.

The GPU filter (vq_vae_amd/csrc/vq_filter.hip) keeps, per latent, the codes whose bf16-based lower
bound L[n,k] lies under a threshold, and only those are evaluated in fp32.  This test rebuilds the
quantities of that rule -- centring on the codebook mean, bf16 rounding, the per-code and per-row
terms of the bound, the threshold -- with the accumulation done in float64 (so that only the
*mathematics* of the bound is under test, not a particular matrix core), and checks on several
data sets that

  * |A[n,k] - (D[n,k] + shift_n)| <= delta[n,k] for every pair, where A is the bf16 contraction,
    D the true squared distance and delta the bound the kernel folds into the contraction;
  * the true argmin (and every code within the fp32 chain's own rounding of it) satisfies
    L[n,k] <= L[n,k^] + 2 (delta[n,k^] - const_n) + 2 const_n, i.e. is a candidate;
  * keys stay positive.
"""
import numpy as np
import pytest

from inputs import gauss

f32 = np.float32


def bf16_rn(x):
    u = np.asarray(x, f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(f32)


def bf16_up(x):
    u = np.asarray(x, f32).view(np.uint32).astype(np.uint64)
    return (((u + 0xFFFF) >> 16) << 16).astype(np.uint32).view(f32)


def bound_terms(z, E):
    """Returns the per-row / per-code quantities exactly as the kernels define them."""
    K, D = E.shape
    mu = (E.astype(np.float64).mean(0)).astype(f32)
    e1 = (E - mu).astype(f32)                      # fl(e - mu)
    z1 = (z - mu).astype(f32)
    et, zt = bf16_rn(e1), bf16_rn(z1)
    # per code
    de = np.sqrt(((e1 - et).astype(np.float64) ** 2).sum(1))
    en = np.sqrt((e1.astype(np.float64) ** 2).sum(1))
    eu = np.sqrt((E.astype(np.float64) ** 2).sum(1))
    pde, pen, pe2 = bf16_up(f32(de * 1.0001)), bf16_up(f32(en * 1.0001)), bf16_up(f32(en ** 2 * 1.0001))
    pun, pu2 = bf16_up(f32(eu * 1.0001)), bf16_up(f32(eu ** 2 * 1.0001))
    n2 = (e1.astype(np.float64) ** 2).sum(1).astype(f32)          # what enters the contraction as ||e'||^2
    # per row
    nz = f32(np.sqrt((zt.astype(np.float64) ** 2).sum(1)) * 1.0001)
    ndz = f32(np.sqrt(((z1 - zt).astype(np.float64) ** 2).sum(1)) * 1.0001)
    zc = nz + ndz
    zn = f32(np.sqrt((z.astype(np.float64) ** 2).sum(1)) * 1.0001)
    gf = f32((D + 24) * 2.0 ** -21 * 1.01)
    go = f32((D + 4) * 2.0 ** -24 * 1.01)
    c1 = bf16_up(2 * nz * f32(1.0001)); c2 = bf16_up((2 * ndz + f32(2.02) * gf * zc) * f32(1.0001))
    c3 = np.full_like(nz, bf16_up(f32(1.01) * gf)); c4 = bf16_up(2 * go * zn * f32(1.0001))
    c5 = np.full_like(nz, bf16_up(go * f32(1.0001)))
    wmax = (c1 * pde.max() + c2 * pen.max() + c3 * pe2.max() + c4 * pun.max() + c5 * pu2.max()) * f32(1.0001)
    cst = (gf * (f32(1.01) * zc * zc + f32(3.1) * wmax) + go * zn * zn * f32(1.0001) + f32(1e-30) + nz * f32(1e-35)) * f32(1.001)
    cup = bf16_up(zc * zc * f32(1.0001) + (2 * wmax + cst) * f32(1.0001))
    return dict(mu=mu, e1=e1, z1=z1, et=et, zt=zt, pde=pde, pen=pen, pe2=pe2, pun=pun, pu2=pu2, n2=n2,
                c=(c1, c2, c3, c4, c5), cst=cst, cup=cup, gf=gf, go=go, zn=zn, eu=eu)


CASES = {
    "gauss": lambda: (gauss(700, (96, 64)), gauss(701, (300, 64))),
    "offset": lambda: (gauss(702, (96, 64)) + f32(7), gauss(703, (300, 64)) + f32(7)),
    "small codes": lambda: (gauss(704, (96, 128)), gauss(705, (200, 128)) * f32(0.05)),
    "wide range": lambda: (gauss(706, (64, 64)) * np.exp(2 * gauss(707, (64, 1))), gauss(708, (256, 64)) * np.exp(gauss(709, (256, 1)))),
    "clustered": lambda: ((gauss(711, (300, 64))[np.arange(96) % 300] + f32(0.01) * gauss(710, (96, 64))).astype(f32),
                          gauss(711, (300, 64))),
}


@pytest.mark.parametrize("name", list(CASES))
def test_bound_holds_and_winner_is_a_candidate(name):
    z, E = CASES[name]()
    z, E = z.astype(f32), E.astype(f32)
    t = bound_terms(z, E)
    c1, c2, c3, c4, c5 = t["c"]
    z64, E64 = z.astype(np.float64), E.astype(np.float64)
    Dtrue = ((z64[:, None, :] - E64[None, :, :]) ** 2).sum(2)                       # [N, K] real distances
    # bf16 contraction in exact arithmetic: c~ + ||e'||^2 - 2 z~.e~
    A = t["cup"].astype(np.float64)[:, None] + t["n2"].astype(np.float64)[None, :] \
        - 2 * t["zt"].astype(np.float64) @ t["et"].astype(np.float64).T
    W = (c1[:, None] * t["pde"][None, :] + c2[:, None] * t["pen"][None, :] + c3[:, None] * t["pe2"][None, :]
         + c4[:, None] * t["pun"][None, :] + c5[:, None] * t["pu2"][None, :]).astype(np.float64)
    delta = t["cst"].astype(np.float64)[:, None] + W
    shift = t["cup"].astype(np.float64) - (t["z1"].astype(np.float64) ** 2).sum(1)
    err = np.abs(A - (Dtrue + shift[:, None]))
    # the filter-side part of delta alone must cover the bf16 and centring errors (the fp32-chain part
    # go (||z|| + ||e||)^2 is reserved for the chain's own rounding)
    chain = t["go"].astype(np.float64) * (t["zn"].astype(np.float64)[:, None] + t["eu"][None, :]) ** 2
    assert (err <= delta - chain).all(), float((err / (delta - chain)).max())
    L = A - W                                                                       # what the keys hold
    assert (L > 0).all()
    kbest = L.argmin(1)
    rows = np.arange(len(z))
    thr = L[rows, kbest] * (1 + 2.0 ** -10) + 2 * (W[rows, kbest] + t["cst"].astype(np.float64))
    cand = L <= thr[:, None]
    dmin = Dtrue.min(1)
    near = Dtrue <= dmin[:, None] + chain.min(1)[:, None]            # codes the fp32 chain could still prefer
    assert (cand | ~near).all()
    assert cand[rows, Dtrue.argmin(1)].all()
    # the rule is selective on separable data, not only safe
    if name in ("gauss", "offset", "clustered"):
        assert cand.sum(1).mean() < 4.0, cand.sum(1).mean()
