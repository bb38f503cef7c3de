import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops, stats, EMAVectorQuantizer
dev = torch.device('cuda:0'); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
# ---- config 3: statistics over 10 M latents
K, D = 8192, 256
E = torch.randn(K, D, device=dev, generator=g)
m = EMAVectorQuantizer(K, D, 3.0).to(dev)
with torch.no_grad(): m.embedding.weight.copy_(E)
batches = [torch.randn(1000, D, 32, 32, device=dev, generator=g) for _ in range(2)]   # 1.024 M latents each
for mode in ('fp32', 'filtered'):
    m.search_mode = mode
    def run():
        return stats.analyze_codebook_utilization(m, (batches[i % 2] for i in range(10)))
    t = timeit(run, reps=2, warm=1)
    r = run()
    print(f'config 3, analyze_codebook_utilization over 10.24 M latents (K=8192, D=256), {mode}: {t:.1f} ms  utilization {r["utilization"]:.3f} entropy {r["entropy"]:.3f}', flush=True)
idx = torch.randint(0, K, (10_000_000,), device=dev, generator=g)
print(f'config 3, index statistics of 10 M int64 indices: {timeit(lambda: stats.analyze_indices(idx, K)):.3f} ms')
for Kp in (8192, 65536):
    Ep = torch.randn(Kp, D, device=dev, generator=g)
    print(f'config 3, pairwise codebook statistics K={Kp}: {timeit(lambda: stats.pairwise_codebook_stats(Ep)):.2f} ms')
# ---- config 5: one shard (8192 rows) of a K=65536 codebook, per-shard search with code_base
Es = torch.randn(8192, D, device=dev, generator=g)
for nb in (512, 4096):
    z = torch.randn(nb, D, 32, 32, device=dev, generator=g)
    for filt in (False, True):
        t = timeit(lambda: ops.pack_dist_index(*(lambda r: (r['dmin'], r['indices']))(ops.search(z, Es, ws, code_base=3 * 8192, want_quantized=False, want_dmin=True, filtered=filt))), reps=3, warm=1)
        print(f'config 5, shard search + pack, N={nb*1024}, filtered={filt}: {t:.2f} ms', flush=True)
    del z
# whole K=65536 on one GPU, N = 512 Ki
E65 = torch.randn(65536, D, device=dev, generator=g)
z = torch.randn(512, D, 32, 32, device=dev, generator=g)
for filt in (False, True):
    print(f'K=65536 on one GPU, N=524288, filtered={filt}: {timeit(lambda: ops.search(z, E65, ws, filtered=filt), reps=3, warm=1):.2f} ms', flush=True)
a = ops.search(z, E65, ws)['indices'].clone(); b = ops.search(z, E65, ws, filtered=True)['indices']
print('K=65536 indices equal:', bool(torch.equal(a, b)))
# ---- config 4: the module inside a training step at a per-GPU batch of 64 images (64 Ki latents):
# forward + backward of the quantizer, then the same with the convolutions either side (QuantizerBoundary)
from vq_vae_amd import QuantizerBoundary
del z, E65
h = torch.randn(64, D, 32, 32, device=dev, generator=g)
wt = torch.randn(64, D, 32, 32, device=dev, generator=g)
for mode in ('fp32', 'filtered'):
    q = EMAVectorQuantizer(K, D, 3.0).to(dev)
    with torch.no_grad(): q.embedding.weight.copy_(E)
    q.search_mode = mode; q.train()
    hh = h.clone().requires_grad_(True)
    def fb():
        out, loss, _ = q(hh)
        ((out * wt).sum() + loss).backward()
        hh.grad = None
    with torch.no_grad():
        t_f = timeit(lambda: q(h), reps=20, warm=3)
    t_fb = timeit(fb, reps=20, warm=3)
    print(f'config 4, module at 64 Ki latents, {mode}: forward {t_f:.3f} ms, forward+backward {t_fb:.3f} ms', flush=True)
    b = QuantizerBoundary(D, K, D).to(dev)
    with torch.no_grad(): b.quantize.embedding.weight.copy_(E)
    b.quantize.search_mode = mode; b.train()
    def fb_fused():
        out, loss, _ = b(hh)
        ((out * wt).sum() + loss).backward()
        hh.grad = None; b.zero_grad(set_to_none=True)
    def fb_unfused():
        qq, loss, _ = b.quantize(b.to_latents(hh))
        out = b.post_quant_conv(qq)
        ((out * wt).sum() + loss).backward()
        hh.grad = None; b.zero_grad(set_to_none=True)
    print(f'config 4, quant_conv -> quantizer -> post_quant_conv, {mode}: forward+backward {timeit(fb_unfused, reps=10, warm=3):.3f} ms with torch convolutions, {timeit(fb_fused, reps=10, warm=3):.3f} ms folded', flush=True)
