import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0")
C = 256
x = torch.randn(4096, C, 32, 32, device=dev)
W = torch.randn(C, C, device=dev) * 0.06
b = torch.randn(C, device=dev)
ws = ops.Workspace(dev)
for cl in (False, True):
    xx = x.contiguous(memory_format=torch.channels_last) if cl else x
    for _ in range(3):
        ops.conv1x1(xx, W, b, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv1x1(xx, W, b, ws)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("VQ_VAE_AMD_LIB", "default"), "channels_last" if cl else "nchw", "%.3f ms" % (e0.elapsed_time(e1) / 10), flush=True)
