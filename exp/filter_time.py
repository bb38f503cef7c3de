import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
N, K, D = 1 << 22, 8192, 256
z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g); E = torch.randn(K, D, device=dev, generator=g)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    print(os.environ.get("VQ_VAE_AMD_LIB", "default")[-14:], "filtered search 4 Mi rows:", timed(lambda: ops.search(z, E, ws, filtered=True)), "ms", flush=True)
