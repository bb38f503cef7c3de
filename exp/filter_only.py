import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device('cuda:0'); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
z = torch.randn(4096, 256, 32, 32, device=dev, generator=g)
E = torch.randn(8192, 256, device=dev, generator=g)
for _ in range(5):
    ops.search(z, E, ws, want_quantized=False, filtered=True)
torch.cuda.synchronize()
