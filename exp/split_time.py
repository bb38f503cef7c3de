import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0")
K, D = 8192, 256
E = torch.randn(K, D, device=dev)
ws = ops.Workspace(dev)
for imgs in (33, 64, 65, 96, 128, 160, 196, 256, 391):
    z = torch.randn(imgs, D, 32, 32, device=dev)
    res = []
    for split in (False, True):
        for _ in range(2):
            ops.search(z, E, ws, want_quantized=True, allow_split=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.search(z, E, ws, want_quantized=True, allow_split=split)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print(f"{imgs * 1024:7d} rows ({imgs * 4:4d} row tiles): undivided {res[0]:.3f} ms, with the split plan {res[1]:.3f} ms", flush=True)
