"""vq_pairwise_stats_static (the entry point that allocates nothing) against vq_pairwise_stats,
through the C ABI, twice in a row, the second call on a smaller codebook."""
import numpy as np
import pytest
import torch

from inputs import gauss
from vq_vae_amd import ops
from vq_vae_amd._lib import load

pytestmark = pytest.mark.gpu


def _both(E):
    lib = load()
    K, D = E.shape
    dev = E.device
    s = ops._stream(dev)
    res = []
    for static in (False, True):
        ws = ops.Workspace(dev)
        image, e2, control = ops.prepare_codebook(E, ws)
        nn_dist = torch.full((K,), -1.0, dtype=torch.float32, device=dev)
        nn_idx = torch.full((K,), -1, dtype=torch.int64, device=dev)
        out = torch.full((3,), -1.0, dtype=torch.float64, device=dev)
        if static:
            rc = lib.vq_pairwise_stats_static(ops._ptr(E), ops._ptr(image), ops._ptr(e2), K, D, ops._ptr(nn_dist),
                                              ops._ptr(nn_idx), ops._ptr(out), ops._ptr(control), s)
        else:
            blocks = lib.vq_search_blocks(K)
            sp = torch.empty(blocks, dtype=torch.float64, device=dev)
            mm = torch.empty(2 * blocks + K, dtype=torch.float32, device=dev)
            rc = lib.vq_pairwise_stats(ops._ptr(E), ops._ptr(image), ops._ptr(e2), K, D, ops._ptr(nn_dist),
                                       ops._ptr(nn_idx), ops._ptr(sp), ops._ptr(mm), ops._ptr(out),
                                       ops._ptr(control), s)
        assert rc == 0, lib.vq_status_string(rc)
        torch.cuda.synchronize()
        res.append((nn_dist.cpu().numpy(), nn_idx.cpu().numpy(), out.cpu().numpy()))
    return res


def test_static_pairwise_equals_workspace_version_twice():
    dev = torch.device("cuda:0")
    for seed, K, D in ((901, 4096, 64), (902, 1000, 64)):      # second call: smaller, ragged K
        E = torch.from_numpy(gauss(seed, (K, D))).to(dev)
        (d0, i0, o0), (d1, i1, o1) = _both(E)
        assert np.array_equal(i0, i1)
        assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
        assert np.array_equal(o0.view(np.uint64), o1.view(np.uint64))
        assert (i1 >= 0).all() and np.isfinite(o1).all()
        # and against float64 on the host
        Ed = E.double().cpu().numpy()
        P = ((Ed[:, None, :] - Ed[None, :, :]) ** 2).sum(-1) if K <= 1024 else None
        if P is not None:
            np.fill_diagonal(P, np.inf)
            assert np.array_equal(P.argmin(1), i1)
            np.testing.assert_allclose(P.min(1), d1, rtol=1e-4)


def test_static_pairwise_rejects_sizes_beyond_its_maximum():
    lib = load()
    dev = torch.device("cuda:0")
    t = torch.zeros(64, device=dev)
    rc = lib.vq_pairwise_stats_static(ops._ptr(t), ops._ptr(t), ops._ptr(t), 65537, 64, ops._ptr(t), ops._ptr(t),
                                      ops._ptr(t), ops._ptr(t), ops._stream(dev))
    assert rc == 3
    assert b"maximum" in lib.vq_status_string(rc)


def test_static_pairwise_allocates_and_frees_nothing():
    """The property the entry point exists for: after the first call on the device (which looks the
    scratch array's address up), a call changes neither the device's free memory (hipMemGetInfo, which
    sees every allocation of the process, the library's included) nor torch's allocator."""
    lib = load()
    dev = torch.device("cuda:0")
    assert lib.vq_init(0) == 0 and lib.vq_large_lds_kernels() > 0      # set-up outside the measured call
    K, D = 2048, 64
    E = torch.from_numpy(gauss(903, (K, D))).to(dev)
    ws = ops.Workspace(dev)
    image, e2, control = ops.prepare_codebook(E, ws)
    nn_dist = torch.empty(K, dtype=torch.float32, device=dev)
    nn_idx = torch.empty(K, dtype=torch.int64, device=dev)
    out = torch.empty(3, dtype=torch.float64, device=dev)
    s = ops._stream(dev)
    args = (ops._ptr(E), ops._ptr(image), ops._ptr(e2), K, D, ops._ptr(nn_dist), ops._ptr(nn_idx), ops._ptr(out),
            ops._ptr(control), s)
    assert lib.vq_pairwise_stats_static(*args) == 0                    # first call on this device
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    alloc0, reserved0 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    for _ in range(3):
        assert lib.vq_pairwise_stats_static(*args) == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, (free0, free1)
    assert torch.cuda.memory_allocated(dev) == alloc0 and torch.cuda.memory_reserved(dev) == reserved0
    assert np.isfinite(out.cpu().numpy()).all()
