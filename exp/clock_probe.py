import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0")
C = 256
x = torch.randn(4096, C, 32, 32, device=dev)
W = torch.randn(C, C, device=dev) * 0.06
b = torch.randn(C, device=dev)
ws = ops.Workspace(dev)
for _ in range(4):
    ops.conv1x1(x, W, b, ws)
gy = torch.randn(4096, C, 32, 32, device=dev)
for _ in range(4):
    ops.conv1x1_wgrad(gy, x, ws)
del gy
E2 = torch.randn(8192, C, device=dev)
for _ in range(2):
    ops.search(x[:1024], E2, ws, want_quantized=True)
torch.cuda.synchronize()
