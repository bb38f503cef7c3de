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
