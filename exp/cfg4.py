import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import ops
training = sys.argv[1] == "train"
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
K, D = 8192, 256
z = torch.randn(64, D, 32, 32, device=dev, generator=g); E = torch.randn(K, D, device=dev, generator=g)
w = torch.zeros(K, D, device=dev); c = torch.zeros(K, device=dev); u = torch.zeros(K, device=dev)
ws = ops.Workspace(dev)
def step(): ops.quantize_step(z, E, w, c, u, beta=3.0, decay=0.95, epsilon=1e-5, temperature=1.0, training=training, ws=ws, filtered=True)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print(sys.argv[1], (time.perf_counter() - t0) / 100 * 1e3, "ms per step; open rows", ws._bufs["control"][:64].view(torch.int32)[8:16].tolist() if ws.clean else ws._bufs["control"][:64].view(torch.int32)[:8].tolist())
