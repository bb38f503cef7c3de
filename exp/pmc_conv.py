import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0")
C = 256
rows = 1024 * 1024
x = torch.randn(rows // 1024, C, 32, 32, device=dev)
W = torch.randn(C, C, device=dev) * 0.06
b = torch.randn(C, device=dev)
ws = ops.Workspace(dev)
for _ in range(3):
    ops.conv1x1(x, W, b, ws)
E = torch.randn(256, C, device=dev)      # a 256-row codebook: the same sweep length as the convolution
for _ in range(3):
    ops.search(x, E, ws, want_quantized=True)
E2 = torch.randn(8192, C, device=dev)
z2 = x[:64]
for _ in range(3):
    ops.search(z2, E2, ws, want_quantized=True, allow_split=False)
torch.cuda.synchronize()
