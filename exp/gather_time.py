import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
N, K, D = 1 << 22, 8192, 256
z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g); E = torch.randn(K, D, device=dev, generator=g)
idx = torch.randint(0, K, (N,), device=dev, generator=g)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("gather_st NCHW 4 Mi rows:", timed(lambda: ops.gather_st(z, E, idx, ws)), "ms")
zr = z.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
print("gather_st rows 4 Mi rows:", timed(lambda: ops.gather_st(zr, E, idx, ws)), "ms")
