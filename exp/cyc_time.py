import ctypes, torch, sys, os
from ctypes import c_void_p as vp, c_int as i32, c_int64 as i64, c_float as f32
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device('cuda:0')
D, B, H = int(sys.argv[2]) if len(sys.argv) > 2 else 256, 1024, 32
N = B*H*H
z = torch.randn(B, D, H, H, device=dev)
l = ctypes.CDLL(os.path.join(here, f'libsearch_{sys.argv[1]}.so'))
l.vq_image_bytes.restype = ctypes.c_size_t; l.vq_image_bytes.argtypes=[i32,i32]
l.vq_loss_partial_bytes.restype = ctypes.c_size_t; l.vq_loss_partial_bytes.argtypes=[i64]
l.vq_control_reset.argtypes=[vp,vp]
l.vq_codebook_prep.argtypes=[vp,i32,i32,vp,vp,vp,vp]
l.vq_search.argtypes=[vp,i32,i64,i32,i64,vp,vp,vp,i32,f32,i64,vp,vp,vp,vp,vp,vp,vp,vp,vp]
for K in (64, 512, 2048):
    E = torch.randn(K, D, device=dev)
    image = torch.empty(l.vq_image_bytes(K,D), dtype=torch.uint8, device=dev)
    e2 = torch.empty(K+64, device=dev); control = torch.zeros(16, dtype=torch.int32, device=dev)
    idx64 = torch.empty(N, dtype=torch.int64, device=dev); idx32 = torch.empty(N, dtype=torch.int32, device=dev)
    nf = torch.empty(N, dtype=torch.int32, device=dev); q = torch.empty_like(z); lp = torch.empty(l.vq_loss_partial_bytes(N)//4, device=dev)
    dbg = torch.zeros(2048*9, device=dev)
    p = lambda t: vp(t.data_ptr())
    l.vq_control_reset(p(control), None); l.vq_codebook_prep(p(E),K,D,p(image),p(e2),p(control),None)
    for _ in range(3):
        rc = l.vq_search(p(z),1,N,D,H*H,p(E),p(image),p(e2),K,1.0,0,p(idx64),p(idx32),None,p(q),p(lp),p(nf),p(control),p(dbg),None)
    torch.cuda.synchronize()
    d = dbg.view(2048, 9).double()
    for grp, sl in (('waves0-3', [w for w in range(2048) if (w % 8) < 4]), ('waves4-7', [w for w in range(2048) if (w % 8) >= 4])):
        m = d[sl].mean(0) / 1e3   # kilocycles
        print(f'K={K} {grp}: work idle/mem/sweep {m[0]:.3f} {m[1]:.3f} {m[2]:.3f} ms | barrier wait {m[3]:.3f} {m[4]:.3f} {m[5]:.3f} | end-of-step waitcnt {m[6]:.3f} {m[7]:.3f} {m[8]:.3f} kcycles', flush=True)
