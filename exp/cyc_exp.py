import ctypes, torch, sys, os
from ctypes import c_void_p as vp, c_int as i32, c_int64 as i64, c_float as f32
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device('cuda:0')
D, B, H = 256, 1024, 32
N = B*H*H
z = torch.randn(B, D, H, H, device=dev)
variants = sys.argv[1:]
for K in (64, 512, 2048):
  E = torch.randn(K, D, device=dev)
  for v in variants:
    name, _, mode = v.partition(':')
    l = ctypes.CDLL(os.path.join(here, f'libsearch_{name}.so'))
    l.vq_image_bytes.restype = ctypes.c_size_t; l.vq_image_bytes.argtypes=[i32,i32]
    l.vq_loss_partial_bytes.restype = ctypes.c_size_t; l.vq_loss_partial_bytes.argtypes=[i64]
    l.vq_control_reset.argtypes=[vp,vp]
    l.vq_codebook_prep.argtypes=[vp,i32,i32,vp,vp,vp,vp]
    l.vq_search.argtypes=[vp,i32,i64,i32,i64,vp,vp,vp,i32,f32,i64,vp,vp,vp,vp,vp,vp,vp,vp,vp]
    l.vq_internal_tune(0, 0 if mode == 'old' else 1)
    image = torch.empty(l.vq_image_bytes(K,D), dtype=torch.uint8, device=dev)
    e2 = torch.empty(K+64, device=dev); control = torch.zeros(16, dtype=torch.int32, device=dev)
    idx64 = torch.empty(N, dtype=torch.int64, device=dev); idx32 = torch.empty(N, dtype=torch.int32, device=dev)
    nf = torch.empty(N, dtype=torch.int32, device=dev); q = torch.empty_like(z); lp = torch.empty(l.vq_loss_partial_bytes(N)//4, device=dev)
    p = lambda t: vp(t.data_ptr())
    l.vq_control_reset(p(control), None); l.vq_codebook_prep(p(E),K,D,p(image),p(e2),p(control),None)
    def run():
        return l.vq_search(p(z),1,N,D,H*H,p(E),p(image),p(e2),K,1.0,0,p(idx64),p(idx32),None,p(q),p(lp),p(nf),p(control),None,None)
    run(); torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for it in range(5): rc = run()
      b.record(); torch.cuda.synchronize()
      best = min(best, a.elapsed_time(b)/5)
    print(f'K={K} {v}: rc {rc} ms {best:.3f} TF {2*N*K*D/best/1e9:.1f}', flush=True)
