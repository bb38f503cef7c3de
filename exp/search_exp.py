import ctypes, torch, sys, os
from ctypes import c_void_p as vp, c_int as i32, c_int64 as i64, c_float as f32
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device('cuda:0')
K, D, B, H = 8192, 256, 4096, 32
N = B*H*H
z = torch.randn(B, D, H, H, device=dev)
E = torch.randn(K, D, device=dev)
variants = sys.argv[1:]
res = {}
ref_idx = None
for rnd in range(3):
  for v in variants:
    l = ctypes.CDLL(os.path.join(here, f'libsearch_{v}.so'))
    l.vq_image_bytes.restype = ctypes.c_size_t; l.vq_image_bytes.argtypes=[i32,i32]
    l.vq_control_reset.argtypes=[vp,vp]
    l.vq_codebook_prep.argtypes=[vp,i32,i32,vp,vp,vp,vp]
    l.vq_search.argtypes=[vp,i32,i64,i32,i64,vp,vp,vp,i32,f32,i64,vp,vp,vp,vp,vp,vp,vp,vp,vp]
    image = torch.empty(l.vq_image_bytes(K,D), dtype=torch.uint8, device=dev)
    e2 = torch.empty(K+64, device=dev); control = torch.zeros(16, dtype=torch.int32, device=dev)
    idx64 = torch.empty(N, dtype=torch.int64, device=dev); idx32 = torch.empty(N, dtype=torch.int32, device=dev)
    nf = torch.empty(N, dtype=torch.int32, device=dev); q = torch.empty_like(z); lp = torch.empty(N//256+1024, device=dev)
    p = lambda t: vp(t.data_ptr())
    l.vq_control_reset(p(control), None); l.vq_codebook_prep(p(E),K,D,p(image),p(e2),p(control),None)
    def run():
        return l.vq_search(p(z),1,N,D,H*H,p(E),p(image),p(e2),K,1.0,0,p(idx64),p(idx32),None,p(q),p(lp),p(nf),p(control),None,None)
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(3): rc = run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)/3
    if ref_idx is None: ref_idx = idx64.clone()
    same = bool(torch.equal(ref_idx, idx64))
    res.setdefault(v, []).append(ms)
    print(rnd, v, 'rc', rc, 'ms %.2f' % ms, 'TF %.1f' % (2*N*K*D/ms/1e9), 'same_idx', same, flush=True)
print({v: round(min(t), 2) for v, t in res.items()})
