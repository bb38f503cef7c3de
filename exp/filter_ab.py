import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
dev = torch.device('cuda:0'); ws = ops.Workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
z = torch.randn(4096, 256, 32, 32, device=dev, generator=g)
E = torch.randn(8192, 256, device=dev, generator=g)
ref = ops.search(z[:256], E, ws, want_quantized=False, filtered=False)["indices"].clone()
got = ops.search(z[:256], E, ws, want_quantized=False, filtered=True)["indices"].clone()
for _ in range(3):
    ops.search(z, E, ws, want_quantized=False, filtered=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    ops.search(z, E, ws, want_quantized=False, filtered=True)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("VQ_VAE_AMD_LIB", "default").split("/")[-1], "filtered search %.3f ms" % (e0.elapsed_time(e1) / 8), "indices equal:", bool(torch.equal(ref, got)), flush=True)
