import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device('cuda:0'); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
for K, D, B in ((8192, 256, 1024), (512, 256, 1024), (64, 256, 1024), (8192, 64, 1024), (512, 64, 1024)):
    z = torch.randn(B, D, 32, 32, device=dev, generator=g); E = torch.randn(K, D, device=dev, generator=g)
    prep = ops.prepare_codebook(E, ws)
    for T in (1.0, 2.5, 1.0, 2.5):
        prep = ops.prepare_codebook(E, ws)
        for _ in range(2): ops.search(z, E, ws, temperature=T, prepared=prep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.search(z, E, ws, temperature=T, prepared=prep)
        e1.record(); torch.cuda.synchronize()
        print(f'K={K} D={D} T={T}: {e0.elapsed_time(e1)/5:.3f} ms, rows sent to the exact kernel {int(prep[2][0])} + {int(prep[2][6])}', flush=True)
