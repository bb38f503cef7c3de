"""The quantizer together with the two 1x1 convolutions either side of it.

Mirrors the segment of the reference's ``CustomVQGAN`` between encoder and decoder
(vqvae/custom_vqgan.py): ``quant_conv`` (:50, applied at :66-68), ``quantize`` (:36-41, :81) and
``post_quant_conv`` (:51, applied at :72), with the same attribute names, so the matching entries of
a ``CustomVQGAN`` state dict load unchanged.

What is fused: the quantizer's output consists of codebook rows, so ``post_quant_conv`` of the N
quantized latents equals a gather from the K convolved codebook rows ``P = E W^T + b``.  The forward
computes P (a K x D x D' product instead of N x D x D'), and one kernel gathers ``P[idx]`` and
accumulates the commitment loss; the N x D tensor of quantized latents is never written or re-read.
The backward reads the quantized row from a saved copy of the codebook (K x D floats instead of
N x D), gets the weight gradient from per-code sums of the incoming gradient (the scatter kernel of
the EMA update) and needs only one N x D x D' product, for the gradient of the latents.

``quant_conv`` cannot be folded the same way: its product must exist in memory as the latents ``z``
(the search, the EMA scatter and the backward all read it), and the search is specified on the fp32
values of ``z``.  It runs as ``vq_conv1x1``, the search kernel's own engine with the convolution's
weights in the codebook's place (as do the two input-gradient products of the backward).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import load
from .ema_vector_quantizer import EMAVectorQuantizer, LazyOneHot


class _QuantizeProjectFn(torch.autograd.Function):
    """forward: search + EMA update + gather of the convolved codebook.  backward: see module text.

    With q = E[idx] (constant w.r.t. everything, as in the reference where the codebook gets no
    gradient and q_st = z + (q - z).detach(), vqvae/ema_vector_quantizer.py:95):
        out      = W q + b
        grad_z   = W^T grad_out + grad_loss * 2 beta / (N D) * (z - q)
        grad_W   = sum_n grad_out[n] q[n]^T = G^T E,  G[k] = sum of grad_out rows assigned to code k
        grad_b   = sum_n grad_out[n] = sum_k G[k]
    """

    @staticmethod
    def forward(ctx, z, weight, bias, module, temperature):
        E = module.embedding.weight.data
        DP, D = weight.shape[0], weight.shape[1]
        W = weight.detach().reshape(DP, D).float()
        needs = ctx.needs_input_grad
        # the EMA update overwrites the codebook in place; the backward needs the rows the search saw
        E_pre = E.clone() if (needs[0] or needs[1] or needs[2]) else E
        P = E @ W.t() if bias is None else torch.addmm(bias.detach().float(), E, W.t())
        res = module._run_step(z, temperature, project=P)
        if res.finish is not None:       # data-parallel mode: the EMA update behind the all-reduce
            res.finish()
        ctx.save_for_backward(z, res.indices, E_pre, weight)
        ctx.has_bias = bias is not None
        ctx.ws = module._workspace(z.device)
        ctx.coef = 2.0 * float(module.commitment_cost) / float(max(z.numel(), 1))
        ctx.mark_non_differentiable(res.perplexity, res.indices)
        return res.quantized, res.loss, res.perplexity, res.indices

    @staticmethod
    def backward(ctx, grad_out, grad_loss, _gp, _gi):
        z, idx, E_pre, weight = ctx.saved_tensors
        DP, D = weight.shape[0], weight.shape[1]
        K = E_pre.shape[0]
        W = weight.detach().reshape(DP, D).float()
        grad_z = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            gq = None
            if grad_out is not None:
                go = grad_out.float()
                if ops.conv1x1_supported(go, D, DP):
                    gq = ops.conv1x1(go, W.t().contiguous(), None, ctx.ws, "post_conv_t_image")
                else:
                    gq = F.conv2d(go, W.t().reshape(D, DP, 1, 1))   # W^T grad_out, per latent
            grad_z = ops.backward_gather(gq, grad_loss, z, E_pre, idx, ctx.coef)
        if grad_out is not None and (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])):
            G = ops.code_sums(grad_out, idx.to(torch.int32), K, ctx.ws)          # (K, D')
            if ctx.needs_input_grad[1]:
                grad_w = (G.t() @ E_pre).reshape(weight.shape).to(weight.dtype)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                grad_b = G.sum(0)
        return grad_z, grad_w, grad_b, None, None


class _Conv1x1Fn(torch.autograd.Function):
    """A 1x1 convolution (vqvae/custom_vqgan.py:50, applied at :68) whose forward and input gradient
    run on the search's MFMA engine (``vq_conv1x1``: the weights stream through LDS like a 256-row
    codebook) and whose weight and bias gradients, sums over all N rows, are ``vq_conv1x1_wgrad``
    (torch's convolution backward only for shapes that kernel does not take)."""

    @staticmethod
    def forward(ctx, x, weight, bias, ws):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.ws = ws
        return ops.conv1x1(x, weight, bias, ws, "quant_conv_image")

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        O, C = weight.shape[0], weight.shape[1]
        grad_x = grad_w = grad_b = None
        go = grad_out.float()
        if ctx.needs_input_grad[0]:
            Wt = weight.detach().reshape(O, C).t().contiguous()
            if ops.conv1x1_supported(go, C, O):
                grad_x = ops.conv1x1(go, Wt, None, ctx.ws, "quant_conv_t_image")
            else:
                grad_x = F.conv2d(go, Wt.reshape(C, O, 1, 1))
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if (want_w or want_b) and ops.conv1x1_wgrad_supported(go, x):
            dW, db = ops.conv1x1_wgrad(go, x, ctx.ws, want_bias=want_b)
            if want_w:
                grad_w = dW.reshape(weight.shape).to(weight.dtype)
            if want_b:
                grad_b = db
        else:
            if want_w:
                grad_w = torch.nn.grad.conv2d_weight(x.float(), weight.shape, go).to(weight.dtype)
            if want_b:
                grad_b = go.sum((0, 2, 3))
        return grad_x, grad_w, grad_b, None


class QuantizerBoundary(nn.Module):
    """quant_conv -> EMAVectorQuantizer -> post_quant_conv (GPU only).

    Constructor arguments as ``CustomVQGAN``'s (vqvae/custom_vqgan.py:17-21), including its check
    that the encoder's channels equal the embedding dimension (:25-26).
    """

    def __init__(self, latent_channels: int = 256, n_embed: int = 8192, embed_dim: int = 256,
                 ema_decay: float = 0.95, commitment_loss_beta: float = 3.0):
        super().__init__()
        if embed_dim != latent_channels:
            raise ValueError(f"Latent channels from encoder ({latent_channels}) must match VQ embedding "
                             f"dimension ({embed_dim}).")
        self.quantize = EMAVectorQuantizer(num_embeddings=n_embed, embedding_dim=embed_dim,
                                           commitment_cost=commitment_loss_beta, decay=ema_decay)
        self.quant_conv = nn.Conv2d(latent_channels, embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, embed_dim, 1)
        self.config = {"num_vq_embeddings": n_embed, "vq_embed_dim": embed_dim, "ema_decay": ema_decay,
                       "commitment_loss_beta": commitment_loss_beta}

    # ------------------------------------------------------------------ the reference's pieces
    def to_latents(self, h: torch.Tensor) -> torch.Tensor:
        """The tail of ``CustomVQGAN.encode`` (:66-68): drop the log-variance half if the encoder
        produced one, then ``quant_conv``."""
        if h.shape[1] == 2 * self.quant_conv.in_channels:
            h, _ = torch.chunk(h, 2, dim=1)
        qc = self.quant_conv
        if h.is_cuda and h.dim() == 4 and ops.conv1x1_supported(h, qc.out_channels, qc.in_channels):
            if not h.is_contiguous() and not h.is_contiguous(memory_format=torch.channels_last):
                h = h.contiguous()
            return _Conv1x1Fn.apply(h, qc.weight, qc.bias, self.quantize._workspace(h.device))
        return qc(h)       # channel counts outside the kernel's set (and the host-side shape checks on CPU)

    def forward(self, h: torch.Tensor, temperature: float = 1.0):
        """h: encoder output (B, C, H, W).  Returns ``(decoder_input, commitment_loss,
        (perplexity, encoding_indices, encodings))`` where ``decoder_input`` is what
        ``CustomVQGAN.decode`` feeds its decoder: post_quant_conv(quantize(quant_conv(h))[0])."""
        if not isinstance(h, torch.Tensor) or not h.is_cuda:
            raise RuntimeError("QuantizerBoundary runs only on the GPU (HIP kernels for MI355X); there is no "
                               f"CPU fallback (got input on '{getattr(h, 'device', type(h))}')")
        load()
        z = self.to_latents(h)
        q = self.quantize
        if z.dim() != 4 or z.shape[1] != q.embedding_dim:
            raise RuntimeError(f"shape '[-1, {q.embedding_dim}]' is invalid for latents of size {tuple(z.shape)}")
        q._follow(z.device)
        if z.dtype != torch.float32:
            z = z.float()
        if not z.is_contiguous() and not z.is_contiguous(memory_format=torch.channels_last):
            z = z.contiguous()
        pc = self.post_quant_conv
        out, loss, perplexity, idx = _QuantizeProjectFn.apply(z, pc.weight, pc.bias, q, float(temperature))
        n = idx.numel()
        if n * q.num_embeddings * 4 <= q.onehot_eager_bytes:
            encodings = ops.onehot(idx, q.num_embeddings)
        else:
            encodings = LazyOneHot(idx, q.num_embeddings)
        return out, loss, (perplexity, idx.view(n, 1), encodings)

    def forward_dict(self, h: torch.Tensor, temperature: float = 1.0) -> dict:
        """The quantizer-side entries of ``CustomVQGAN.forward(return_dict=True)`` (:107-114), all as
        device tensors: nothing is copied to the host (the reference's get_codebook_stats costs four
        host synchronisations per forward, :102-104).  ``indices`` holds the one-hot encodings, as
        the reference's dict does (:92)."""
        out, loss, (perplexity, _idx, enc) = self.forward(h, temperature)
        stats = self.quantize.codebook_stats_device()
        return {"decoder_input": out, "commitment_loss": loss, "indices": enc, "perplexity": perplexity,
                "entropy": stats[2].float(), "normalized_entropy": stats[3].float()}

    # ------------------------------------------------------------------ checkpoints
    def load_vqgan_state_dict(self, vqgan_state_dict: dict, strict: bool = True):
        """Load ``quant_conv.*``, ``post_quant_conv.*`` and ``quantize.*`` out of a ``CustomVQGAN``
        state dict (the ``vqgan_state_dict`` entry of a trainer checkpoint, vqvae/vqgan_trainer.py:409-423);
        encoder / decoder entries are ignored."""
        mine = {k: v for k, v in vqgan_state_dict.items()
                if k.startswith(("quant_conv.", "post_quant_conv.", "quantize."))}
        return self.load_state_dict(mine, strict=strict)
