"""HIP-graph replay of the training step (GraphedQuantizer).

Not collected by a plain `pytest tests` (the file name does not start with test_): run it by name,
    python -m pytest tests/wip_gpu_graph.py -m gpu
It has passed in every run so far (fresh module, training mode, K=512, D=64, 204 replays), but graph
replays outside that scenario ended in GPU memory faults whose cause is not understood (see
vq_vae_amd/graph.py), and a fault takes the whole test process -- and the box -- with it.  Until the
cause is found nothing else should depend on this file's outcome."""
import time

import numpy as np
import pytest
import torch

from inputs import gauss

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make(K, D, beta, decay, E, dev):
    from vq_vae_amd import EMAVectorQuantizer
    m = EMAVectorQuantizer(num_embeddings=K, embedding_dim=D, commitment_cost=beta, decay=decay).to(dev)
    with torch.no_grad():
        m.embedding.weight.copy_(torch.from_numpy(E))
    return m


def test_graphed_step_equals_eager(dev):
    """The captured HIP graph of a training step evolves the module exactly like eager steps."""
    import time
    from vq_vae_amd.graph import GraphedQuantizer
    K, D = 512, 64
    E = gauss(60, (K, D)) * np.float32(0.5)
    batches = [torch.from_numpy(gauss(61 + i, (16, D, 16, 16))).to(dev) for i in range(4)]
    eager = make(K, D, 3.0, 0.95, E, dev); eager.train()
    graphed = make(K, D, 3.0, 0.95, E, dev); graphed.train()
    g = GraphedQuantizer(graphed, batches[0])
    for k, v in graphed.state_dict().items():                          # capture left the state untouched
        assert torch.equal(v, eager.state_dict()[k]), k
    for z in batches:
        q0, l0, (p0, i0, _) = eager(z)
        q1, l1, p1, i1 = g(z)
        assert torch.equal(q0, q1) and torch.equal(i0.view(-1), i1.view(-1))
        assert torch.equal(l0, l1) and torch.equal(p0, p1)
    for k, v in graphed.state_dict().items():
        assert torch.equal(v, eager.state_dict()[k]), k
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g(batches[0])
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        eager(batches[0])
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 200
    print(f"K=512 D=64 N=4096 training step: eager {t_eager * 1e6:.0f} us, graph replay {t_graph * 1e6:.0f} us")
    with pytest.raises(RuntimeError):
        g(torch.zeros(2, D, 4, 4, device=dev))
    # outside the tested envelope the capture refuses; a captured module refuses its own forward
    with pytest.raises(RuntimeError):
        graphed(batches[0])
    used = make(K, D, 3.0, 0.95, E, dev); used.train(); used(batches[0])
    with pytest.raises(RuntimeError):
        GraphedQuantizer(used, batches[0])
    evalm = make(K, D, 3.0, 0.95, E, dev); evalm.eval()
    with pytest.raises(RuntimeError):
        GraphedQuantizer(evalm, batches[0])
    wide = make(64, 256, 3.0, 0.95, gauss(62, (64, 256)), dev); wide.train()
    with pytest.raises(RuntimeError):
        GraphedQuantizer(wide, torch.zeros(2, 256, 4, 4, device=dev))
