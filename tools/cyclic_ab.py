"""The persistent short-sweep search kernel against the one-tile-per-workgroup kernel: same inputs, outputs
compared bit for bit (indices, winning distances, quantized tensor; the loss sum to 1e-6), both timed.
python tools/cyclic_ab.py [rows]      (default 1 Mi latents)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import _lib, ops  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = _lib.load()
    tune = lib.vq_internal_tune
    dev = torch.device("cuda:0")
    ws = ops.Workspace(dev)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    if len(sys.argv) > 2:
        tune(2, int(sys.argv[2]))      # smallest number of tiles per wave the persistent kernel takes
    tune(1, 32)                        # up to 32 stages (K = 2048), to see where it stops paying
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(64, 256), (128, 256), (512, 256), (1024, 256), (2048, 256), (512, 64), (64, 64), (1000, 64), (256, 128),
              (512, 128), (100, 32), (512, 16)]
    print(f"rows {N}: K, D, layout, T: one-tile kernel ms -> persistent ms (TFLOP/s, TB/s of z in + q out) | equal")
    for K, D in shapes:
        for layout in ("nchw", "rows"):
            for T in (1.0, 2.5):
                if (layout == "rows" or T != 1.0) and (K, D) not in ((512, 256), (64, 256), (512, 64)):
                    continue
                z = torch.randn(N // 1024, D, 32, 32, device=dev, generator=g)
                if layout == "rows":
                    z = z.contiguous(memory_format=torch.channels_last)
                E = torch.randn(K, D, device=dev, generator=g)
                res, ms = {}, {}
                for mode in (0, 1):
                    tune(0, mode)
                    r = ops.search(z, E, ws, temperature=T, want_quantized=True, want_dmin=True)
                    res[mode] = (r["indices"].clone(), r["dmin"].clone(), r["quantized"].clone(),
                                 float(r["loss_partial"][: r["n_partial"]].double().sum()))
                    ms[mode] = timed(lambda: ops.search(z, E, ws, temperature=T, want_quantized=True))
                tune(0, 1)
                a, b = res[0], res[1]
                same = (torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
                        and torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
                        and abs(a[3] - b[3]) <= 1e-6 * abs(a[3]))
                fl = 2.0 * N * K * D
                by = 2.0 * N * D * 4
                print(f"{K:5d}, {D:3d}, {layout}, T={T}: {ms[0]:8.3f} -> {ms[1]:8.3f} ms "
                      f"({fl / ms[1] / 1e9:6.1f} TFLOP/s, {by / ms[1] / 1e9:5.2f} TB/s) | {'same' if same else 'DIFFERENT'}",
                      flush=True)
                del z, res


if __name__ == "__main__":
    main()
