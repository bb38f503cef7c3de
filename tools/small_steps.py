"""Step times of the small configurations: configuration 1 (K=512, D=64, 4096 latents) and a per-GPU
training batch of configuration 4 (K=8192, D=256, 64 Ki latents), ops.quantize_step, training mode."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, reps=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, K, D, shape, modes in (('config 1', 512, 64, (16, 64, 16, 16), (False,)), ('config 4 per-GPU batch', 8192, 256, (64, 256, 32, 32), (False, True)),
                                 ('train_vqvae.py default, 16 images', 512, 256, (16, 256, 32, 32), (False,))):
    z = torch.randn(*shape, device=dev, generator=g); E0 = torch.randn(K, D, device=dev, generator=g)
    for filt in modes:
        for training in (True, False):
            E = E0.clone(); w = torch.zeros(K, D, device=dev); c = torch.zeros(K, device=dev); u = torch.zeros(K, device=dev)
            ws = ops.Workspace(dev)
            t = timeit(lambda: ops.quantize_step(z, E, w, c, u, beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=training, ws=ws, filtered=filt))
            print(f'{name}: K={K} D={D} rows={z.numel()//D} filtered={filt} training={training}: {t:.4f} ms per step', flush=True)
