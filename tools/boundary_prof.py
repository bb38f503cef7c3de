import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import QuantizerBoundary
dev = torch.device("cuda:0")
K, D, rows = 8192, 256, 4 * 1024 * 1024
g = torch.Generator(device=dev).manual_seed(0)
z = torch.randn(rows // 1024, D, 32, 32, device=dev, generator=g)
wt = torch.randn(rows // 1024, D, 32, 32, device=dev, generator=g)
b = QuantizerBoundary(D, K, D).to(dev)
with torch.no_grad():
    b.quantize.embedding.weight.copy_(torch.randn(K, D, device=dev, generator=g))
b.quantize.search_mode = "filtered"
b.train()
hh = z.clone().requires_grad_(True)
for _ in range(3):
    out, loss, _ = b(hh)
    ((out * wt).sum() + loss).backward()
    hh.grad = None
    b.zero_grad(set_to_none=True)
torch.cuda.synchronize()
