import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 1 << 20
for K, D in ((64, 256), (512, 256), (512, 64)):
    z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g)
    for en in (0, 1):
        lib.vq_internal_tune(0, en)
        print(K, D, "persistent" if en else "one-tile", f"{timed(lambda: ops.search(z, E, ws, want_quantized=True)):.3f} ms", flush=True)
