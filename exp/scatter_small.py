import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0"); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
K, D = 8192, 256
for images in (4, 4096):
    z = torch.randn(images, D, 32, 32, device=dev, generator=g)
    idx = torch.randint(0, K, (images * 1024,), device=dev, generator=g, dtype=torch.int32)
    ops.embed_sums(z, idx, K, ws)
    for wgs in (128, 256, 512, 1024):
        lib.vq_internal_scatter_wgs(wgs)
        def run():
            slabs, n = ops.embed_sums(z, idx, K, ws, zmax_valid=True)
            return ops.reduce_slabs(slabs)
        t = timed(run)
        nslab = lib.vq_scatter_slabs_for(images * 1024, K, D, 1, 1024)
        print(f"{images} images, aim {wgs} workgroups -> {nslab} slabs: scatter + reduce {t*1e3:.1f} us", flush=True)
lib.vq_internal_scatter_wgs(512)
