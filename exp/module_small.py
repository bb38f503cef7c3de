import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import EMAVectorQuantizer, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, reps=300, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for K, D, shape in ((512, 64, (16, 64, 16, 16)), (512, 256, (16, 256, 32, 32)), (64, 256, (16, 256, 32, 32))):
    m = EMAVectorQuantizer(K, D, 3.0).to(dev)
    with torch.no_grad(): m.embedding.weight.copy_(torch.randn(K, D, device=dev, generator=g))
    z = torch.randn(*shape, device=dev, generator=g)
    m.train()
    with torch.no_grad():
        t_ng = timeit(lambda: m(z))
    zz = z.clone().requires_grad_(True)
    def fb():
        q, loss, _ = m(zz)
        (q.sum() + loss).backward()
        zz.grad = None
    t_fb = timeit(fb)
    m.eval()
    with torch.no_grad():
        t_ev = timeit(lambda: m(z))
    ws = ops.Workspace(dev)
    E = m.embedding.weight.data; w = m.ema_w; c = m.ema_cluster_size; u = m.usage_count
    t_ops = timeit(lambda: ops.quantize_step(z, E, w, c, u, beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=True, ws=ws))
    print(f"K={K} D={D} rows={z.numel()//D}: module train no_grad {t_ng*1e3:.1f} us, train fwd+bwd {t_fb*1e3:.1f} us, eval {t_ev*1e3:.1f} us; ops.quantize_step {t_ops*1e3:.1f} us", flush=True)
