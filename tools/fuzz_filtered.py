"""Fuzz of the filtered search against the fp32 search: random shapes, scales, structures and temperatures;
every index, winning distance and output value must be bit-identical.  python tools/fuzz_filtered.py [seed] [trials]"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device('cuda:0'); ws = ops.Workspace(dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator(device=dev).manual_seed(123)
bad = 0; stats = []
for t in range(trials):
    D = int(rng.choice([64, 128, 256])); K = int(rng.choice([100, 512, 1000, 4096, 8192])); nb = int(rng.choice([1, 3, 16, 64, 200]))
    kind = rng.choice(['gauss', 'scaled', 'heavy', 'lowrank', 'clustered', 'neardup', 'chanscale', 'offset', 'tiny', 'huge'])
    E = torch.randn(K, D, device=dev, generator=g); z = torch.randn(nb, D, 32, 32, device=dev, generator=g)
    if kind == 'scaled':
        E *= float(10 ** rng.uniform(-3, 3)); z *= float(10 ** rng.uniform(-3, 3))
    elif kind == 'heavy':
        z = z * torch.exp(2 * torch.randn(nb, 1, 32, 32, device=dev, generator=g)); E = E * torch.exp(torch.randn(K, 1, device=dev, generator=g))
    elif kind == 'lowrank':
        B = torch.randn(8, D, device=dev, generator=g); E = torch.randn(K, 8, device=dev, generator=g) @ B
        z = (torch.randn(nb * 1024, 8, device=dev, generator=g) @ B + 0.01 * torch.randn(nb * 1024, D, device=dev, generator=g)).reshape(nb, 32, 32, D).permute(0, 3, 1, 2).contiguous()
    elif kind == 'clustered':
        idx = torch.randint(0, K, (nb * 1024,), device=dev, generator=g)
        z = (E[idx] + float(10 ** rng.uniform(-3, 0)) * torch.randn(nb * 1024, D, device=dev, generator=g)).reshape(nb, 32, 32, D).permute(0, 3, 1, 2).contiguous()
    elif kind == 'neardup':
        E[K // 2:] = E[: K - K // 2] + 1e-4 * torch.randn(K - K // 2, D, device=dev, generator=g)
    elif kind == 'chanscale':
        s = torch.exp(3 * torch.randn(D, device=dev, generator=g)); E = E * s; z = z * s.view(1, D, 1, 1)
    elif kind == 'offset':
        mu = 5 * torch.randn(D, device=dev, generator=g); E = E + mu; z = z + mu.view(1, D, 1, 1)
    elif kind == 'tiny':
        E *= 1e-12; z *= 1e-12
    elif kind == 'huge':
        E *= 1e12; z *= 1e12
    T = float(rng.choice([1.0, 1.0, 2.5, 0.3]))
    wd = bool(rng.integers(0, 2))
    a = ops.search(z, E, ws, temperature=T, want_dmin=True)
    ai, ad, aq = a['indices'].clone(), a['dmin'].clone(), a['quantized'].clone()
    b = ops.search(z, E, ws, temperature=T, want_dmin=wd, filtered=True)
    if not wd: b['dmin'] = ad
    torch.cuda.synchronize()
    mi = int((ai != b['indices']).sum()); md = int((ad.view(torch.int32) != b['dmin'].view(torch.int32)).sum()); mq = int((aq.view(torch.int32) != b['quantized'].view(torch.int32)).sum())
    ctl = b['control'][:6].tolist(); N = nb * 1024
    if mi or md or mq:
        bad += 1
        print(f'MISMATCH trial {t}: {kind} K={K} D={D} N={N} T={T}: idx {mi} dmin {md} q {mq} ctl {ctl}', flush=True)
    stats.append((kind, ctl[0] / N, ctl[4] / N, ctl[5] / N))
print(f'{trials} trials, {bad} with mismatches')
for kind in sorted(set(s[0] for s in stats)):
    ss = [s for s in stats if s[0] == kind]
    print(f'  {kind:10s} n={len(ss):3d}  exact-rows {np.mean([s[1] for s in ss]):.4f}  level2 {np.mean([s[2] for s in ss]):.4f} (max {np.max([s[2] for s in ss]):.3f})  hard {np.mean([s[3] for s in ss]):.5f}')
