#!/usr/bin/env python3
"""Configuration 4's data-parallel quantizer step on one rank, for a kernel trace and for timing.

    python3 tools/dp_step_trace.py [steps]
    rocprofv3 --kernel-trace --stats ... -- python3 tools/dp_step_trace.py 20

One process, a one-rank RCCL group (the collective still runs: RCCL launches its kernel), the module at
K = 8192, D = 256 with enable_data_parallel(), a per-GPU batch of 64 images of 32 x 32 latents, training mode.
Under rocprofv3 the kernel table of this run shows what a step launches between the search and the codebook
update: the library's pack / finish kernels and RCCL's all-reduce, and no torch elementwise kernels.
Prints the step time with the exchange and, for comparison, without it (the same module before
enable_data_parallel()).
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_vae_amd import EMAVectorQuantizer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    K, D = 8192, 256
    g = torch.Generator(device=dev).manual_seed(0)
    E = torch.randn(K, D, device=dev, generator=g)
    h = torch.randn(64, D, 32, 32, device=dev, generator=g)

    def run(m, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                m(h)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for mode in ("fp32", "filtered"):
        m = EMAVectorQuantizer(K, D, 3.0).to(dev)
        with torch.no_grad():
            m.embedding.weight.copy_(E)
        m.search_mode = mode
        m.train()
        run(m, 3)
        t_local = run(m, steps)
        m.enable_data_parallel()
        run(m, 3)
        t_dp = run(m, steps)
        print(f"config 4 step at 64 Ki latents, {mode}: {t_local:.3f} ms alone, {t_dp:.3f} ms with the statistics "
              f"exchange over a one-rank RCCL group", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
