import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
N, K, D = 1 << 22, 8192, 256
z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g); E = torch.randn(K, D, device=dev, generator=g)
prep = ops.prepare_codebook(E, ws)
for _ in range(3): ops.search(z, E, ws, prepared=prep, want_quantized=True)
torch.cuda.synchronize()
