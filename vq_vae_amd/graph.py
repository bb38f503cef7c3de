"""HIP-graph replay of the quantizer step for launch-bound (small) batches.

A training forward of the quantizer is a dozen short kernels; at the reference's CPU-sized
configuration (K=512, D=64, 4096 latents) their launches, not their run time, set the step time.
``GraphedQuantizer`` captures the whole step -- codebook prep, search, histogram, EMA sums,
scalars, EMA updates -- for one input shape into a HIP graph (torch.cuda.CUDAGraph on ROCm) and
replays it with one launch.  The module's state tensors are updated in place by the replay exactly
as by the eager forward (vqvae/ema_vector_quantizer.py:61-89).
"""
from __future__ import annotations

import torch


class GraphedQuantizer:
    """Captured ``EMAVectorQuantizer`` step for inputs of ``example``'s shape.

    ``__call__(z)`` copies ``z`` into the captured input buffer, replays the graph and returns
    ``(quantized, loss, perplexity, indices)`` -- tensors that the next replay overwrites.
    Temperature is fixed at capture time.  Not available together with the data-parallel hook
    (collectives are not captured).

    The envelope is deliberately the one that is tested on the GPU (and the one where launches, not
    kernels, set the step time): a module in training mode that has not run yet, embedding_dim <= 64
    (every kernel within the default 64 KiB of LDS), the fp32 search, and no eager forward of the same
    module once it is captured.  Outside it the constructor refuses: replays of graphs captured from a
    module that was also run eagerly, or in eval mode, ended in GPU memory faults on ROCm 7.0's
    runtime whose cause was not found, and a replay at embedding_dim = 256 returned too early to have
    run the search kernel.
    """

    def __init__(self, module, example: torch.Tensor, temperature: float = 1.0, warmup: int = 2):
        if not example.is_cuda:
            raise RuntimeError("GraphedQuantizer needs a GPU tensor")
        if module._stats_reduce is not None:
            raise RuntimeError("the data-parallel hook cannot be captured into a graph")
        if not module.training:
            raise RuntimeError("GraphedQuantizer captures the training step only (see the class text)")
        if module._ws is not None:
            raise RuntimeError("GraphedQuantizer needs a module that has not run yet: capture it before any forward")
        if module.embedding_dim > 64 or module.search_mode != "fp32":
            raise RuntimeError("GraphedQuantizer is limited to embedding_dim <= 64 with the fp32 search: the "
                               "launch-bound shapes it is meant and tested for")
        self.module = module
        self.training = module.training
        self.temperature = float(temperature)
        self.static_in = example.detach().clone().contiguous()
        module._follow(example.device)
        # warm-up on a side stream (allocates the work buffers, sets kernel attributes), then
        # restore the state the warm-up steps changed
        saved = {k: v.clone() for k, v in module.state_dict().items()}
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                module._run_step(self.static_in, self.temperature)
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        with torch.no_grad():
            for k, v in module.state_dict().items():
                v.copy_(saved[k])
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self._out = module._run_step(self.static_in, self.temperature)
        module._captured = True     # the module's own forward refuses from now on

    def __call__(self, z: torch.Tensor):
        if z.shape != self.static_in.shape:
            raise RuntimeError(f"captured for inputs of shape {tuple(self.static_in.shape)}, got {tuple(z.shape)}")
        if self.module.training != self.training:
            raise RuntimeError("the module's train/eval mode changed since the capture")
        self.static_in.copy_(z)
        self.graph.replay()
        o = self._out
        return o.quantized, o.loss, o.perplexity, o.indices
