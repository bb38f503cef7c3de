import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 1 << 22
for K, D in ((8192, 256), (512, 256), (8192, 128)):
    z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g)
    E = torch.randn(K, D, device=dev, generator=g)
    prep = ops.prepare_codebook(E, ws)
    for en in (0, 1, 0, 1):
        lib.vq_internal_tune(0, en)
        t = timed(lambda: ops.search(z, E, ws, prepared=prep, want_quantized=True))
        print(f"{os.environ.get('VQ_VAE_AMD_LIB','default')[-16:]} N={N} K={K} D={D} {'persistent' if en else 'one-tile'}: {t:.3f} ms = {2.0*N*K*D/t/1e9:.1f} TFLOP/s", flush=True)
    del z
