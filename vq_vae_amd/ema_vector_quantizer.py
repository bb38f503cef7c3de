"""Drop-in ``EMAVectorQuantizer`` whose forward runs as HIP kernels on MI355X.

Mirrors the reference module (vqvae/ema_vector_quantizer.py): same constructor
(:11), same six ``state_dict`` entries (:21-31), ``forward(inputs, temperature=1.0)``
returning ``(quantized, commitment_loss, (perplexity, encoding_indices, encodings))`` (:108),
and the four maintenance methods (:110, :165, :205, :234).  The forward never builds the
N x K distance matrix; the N x K one-hot ``encodings`` is produced only when a caller looks at
it (eagerly for small problems, lazily above ``onehot_eager_bytes``).

There is no CPU path: tensors must live on the GPU and libvqhip.so must be built.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from ._lib import check, load


class LazyOneHot:
    """Stands in for the reference's dense ``encodings`` [N, K] fp32 matrix
    (vqvae/ema_vector_quantizer.py:54-55) without allocating it.  Any tensor method or torch
    function applied to it materialises the matrix on the GPU first (one write pass)."""

    def __init__(self, indices: torch.Tensor, num_embeddings: int):
        self._idx = indices.view(-1)
        self._K = int(num_embeddings)
        self._dense: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return torch.Size((self._idx.numel(), self._K))

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self._idx.device

    def size(self, dim: Optional[int] = None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def materialize(self) -> torch.Tensor:
        if self._dense is None:
            self._dense = ops.onehot(self._idx, self._K)
        return self._dense

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(a):
            if isinstance(a, LazyOneHot):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(conv(x) for x in a)
            return a
        return func(*conv(args), **{k: conv(v) for k, v in (kwargs or {}).items()})

    def __repr__(self):
        return f"LazyOneHot(shape={tuple(self.shape)}, device={self.device}, materialized={self._dense is not None})"

    # Python looks operators up on the type, not through __getattr__: each one materialises and delegates.
    def __len__(self):
        return self._idx.numel()

    def __getitem__(self, key):
        if isinstance(key, int) and self._dense is None:
            # one row costs K floats, not the whole matrix
            row = torch.zeros(self._K, dtype=torch.float32, device=self._idx.device)
            row[self._idx[key]] = 1.0
            return row
        return self.materialize()[key]

    def __iter__(self):
        return iter(self.materialize())

    def __bool__(self):
        return bool(self.materialize())

    def __neg__(self):
        return -self.materialize()


def _delegate(name):
    def op(self, *args):
        args = tuple(a.materialize() if isinstance(a, LazyOneHot) else a for a in args)
        return getattr(self.materialize(), name)(*args)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "and", "or", "xor"):
    setattr(LazyOneHot, f"__{_n}__", _delegate(f"__{_n}__"))
    setattr(LazyOneHot, f"__r{_n}__", _delegate(f"__r{_n}__"))
for _n in ("eq", "ne", "lt", "le", "gt", "ge", "abs", "invert", "float", "int"):
    setattr(LazyOneHot, f"__{_n}__", _delegate(f"__{_n}__"))
LazyOneHot.__hash__ = object.__hash__
del _n


class _QuantizeFn(torch.autograd.Function):
    """forward: the HIP step.  backward: grad_z = grad_q + grad_loss * 2*beta/(N*D) * (z - q),
    the autograd of vqvae/ema_vector_quantizer.py:92 (commitment term) and :95 (straight-through);
    nothing flows to the codebook."""

    @staticmethod
    def forward(ctx, z: torch.Tensor, module: "EMAVectorQuantizer", temperature: float):
        res = module._run_step(z, temperature)
        module._pending_finish = res.finish     # data-parallel mode: the all-reduce is in flight (see forward)
        ctx.save_for_backward(z, res.quantized)
        ctx.coef = 2.0 * float(module.commitment_cost) / float(max(z.numel(), 1))
        ctx.mark_non_differentiable(res.perplexity, res.indices)
        return res.quantized, res.loss, res.perplexity, res.indices

    @staticmethod
    def backward(ctx, grad_q, grad_loss, _gp, _gi):
        z, q = ctx.saved_tensors
        return ops.backward(grad_q, grad_loss, z, q, ctx.coef), None, None


class EMAVectorQuantizer(nn.Module):
    """Vector quantizer with exponential-moving-average codebook updates (GPU only)."""

    #: above this many bytes the one-hot ``encodings`` output is lazy instead of dense
    onehot_eager_bytes = 1 << 30
    #: "fp32": every distance on the fp32 matrix cores.  "filtered": a bf16 matrix-core pass narrows
    #: each row to the few codes that can be the fp32 argmin (proven error bound) and only those are
    #: evaluated in fp32 -- the same indices, distances and outputs bit for bit, several times faster
    #: on large batches; shapes it does not cover (D not in {64,128,256}, K > 8192) use "fp32".
    search_mode = "fp32"
    #: "filtered" pays from about this many latent vectors and this much work (rows x codes x channels)
    #: upwards (measured crossover: 0.55 against 0.63 ms at 16 Ki rows, K=8192, D=256) -- its pipeline
    #: of kernels has a floor of ~0.15-0.5 ms; below either, the fp32 search runs
    filtered_min_rows = 16384
    filtered_min_work = 1 << 35

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay=0.95, epsilon=1e-5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.decay = decay
        self.epsilon = epsilon

        self.embedding = nn.Embedding(self.num_embeddings, self.embedding_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.num_embeddings, 1.0 / self.num_embeddings)

        self.register_buffer("ema_cluster_size", torch.zeros(num_embeddings))
        self.register_buffer("ema_w", torch.zeros(num_embeddings, self.embedding_dim))
        self.register_buffer("usage_count", torch.zeros(num_embeddings))
        self.register_buffer("last_reset_epoch", torch.zeros(1))
        self.register_buffer("last_low_usage_reset_epoch", torch.zeros(1))

        # not state: work buffers and the optional data-parallel hook
        self._ws: Optional[ops.Workspace] = None
        self._stats_reduce = None

    # ------------------------------------------------------------------ data-parallel mode
    def enable_data_parallel(self, group=None, uneven_rows: bool = False) -> None:
        """Sum the EMA statistics over the ranks of ``group`` (RCCL) before every update."""
        from .distributed import make_stats_reduce
        self._stats_reduce = make_stats_reduce(group, uneven_rows)

    def disable_data_parallel(self) -> None:
        self._stats_reduce = None

    # ------------------------------------------------------------------ forward
    def _workspace(self, device) -> ops.Workspace:
        if self._ws is None or self._ws.device != device:
            self._ws = ops.Workspace(device)
        return self._ws

    def _follow(self, device) -> None:
        """Buffers and codebook follow the input's device, as the reference's forward does
        (vqvae/ema_vector_quantizer.py:39, :66, :70, :83, :88)."""
        if self.embedding.weight.device != device:
            self.embedding.weight.data = self.embedding.weight.data.to(device)
        for name in ("ema_cluster_size", "ema_w", "usage_count"):
            buf = getattr(self, name)
            if buf.device != device:
                setattr(self, name, buf.to(device))

    def _use_filter(self, z: torch.Tensor) -> bool:
        if self.search_mode not in ("fp32", "filtered"):
            raise ValueError(f"search_mode must be 'fp32' or 'filtered', got {self.search_mode!r}")
        rows = z.numel() // self.embedding_dim
        return (self.search_mode == "filtered" and rows >= self.filtered_min_rows
                and rows * self.num_embeddings * self.embedding_dim >= self.filtered_min_work)

    def _run_step(self, z: torch.Tensor, temperature: float, project=None) -> ops.StepResult:
        E = self.embedding.weight.data
        if not E.is_contiguous() or E.dtype != torch.float32:
            raise RuntimeError("embedding.weight must be a contiguous float32 tensor")
        return ops.quantize_step(z, E, self.ema_w, self.ema_cluster_size, self.usage_count,
                                 beta=float(self.commitment_cost), decay=float(self.decay),
                                 epsilon=float(self.epsilon), temperature=float(temperature),
                                 training=self.training, ws=self._workspace(z.device),
                                 stats_reduce=self._stats_reduce, filtered=self._use_filter(z), project=project,
                                 defer_finish=self._stats_reduce is not None)

    def forward(self, inputs, temperature=1.0):
        if not isinstance(inputs, torch.Tensor) or not inputs.is_cuda:
            raise RuntimeError("EMAVectorQuantizer runs only on the GPU (HIP kernels for MI355X); "
                               f"got input on '{getattr(inputs, 'device', type(inputs))}'. There is no CPU fallback.")
        if inputs.dim() != 4 or inputs.shape[1] != self.embedding_dim:
            # the reference fails in .view(-1, embedding_dim) with a RuntimeError (:36)
            raise RuntimeError(f"shape '[-1, {self.embedding_dim}]' is invalid for input of size "
                               f"{tuple(inputs.shape)}: expected (B, {self.embedding_dim}, H, W)")
        load()  # fail loudly here when libvqhip.so is missing
        self._follow(inputs.device)
        z = inputs
        if z.dtype != torch.float32:
            z = z.float()          # the search always runs in fp32, also under autocast
        if not z.is_contiguous() and not z.is_contiguous(memory_format=torch.channels_last):
            z = z.contiguous()     # channels_last is consumed as it is: in memory it is the (N, D) matrix
        quantized, loss, perplexity, idx = _QuantizeFn.apply(z, self, float(temperature))
        n = idx.numel()
        encoding_indices = idx.view(n, 1)
        if n * self.num_embeddings * 4 <= self.onehot_eager_bytes:
            encodings = ops.onehot(idx, self.num_embeddings)
        else:
            encodings = LazyOneHot(idx, self.num_embeddings)
        self._finish_pending()      # data-parallel mode: the one-hot matrix was written beside the all-reduce
        return quantized, loss, (perplexity, encoding_indices, encodings)

    def _finish_pending(self) -> None:
        """Data-parallel mode: enqueue what waits for the all-reduce of the EMA statistics (loss, perplexity,
        the EMA update).  Always done before forward returns."""
        fin, self._pending_finish = getattr(self, "_pending_finish", None), None
        if fin is not None:
            fin()

    # ------------------------------------------------------------------ statistics
    def get_codebook_stats(self):
        """Utilisation / entropy of the EMA cluster sizes (reference :205-232): one kernel and one
        40-byte copy instead of four host syncs."""
        cs = self.ema_cluster_size
        if not cs.is_cuda:
            raise RuntimeError("get_codebook_stats needs the module on the GPU")
        lib = load()
        out = torch.empty(5, dtype=torch.float64, device=cs.device)
        check(lib.vq_cluster_stats(ops._ptr(cs), ops._ptr(self.usage_count.to(cs.device)), self.num_embeddings,
                                   ops._ptr(out), ops._stream(cs.device)), "vq_cluster_stats")
        v = out.cpu().tolist()
        active = int(v[0])
        return {
            "active_size": active,
            "utilization": active / self.num_embeddings,
            "entropy": v[2] if active > 0 else 0,
            "normalized_entropy": v[3] if active > 0 else 0,
            "total_usage": v[4],
        }

    def codebook_stats_device(self) -> torch.Tensor:
        """The numbers of get_codebook_stats as a float64 device tensor [active_size, utilization,
        entropy, normalized_entropy, total_usage] -- no copy to the host, no synchronisation (the
        reference reads them with four .item() calls on every forward that returns a dict,
        vqvae/custom_vqgan.py:102-104).  Entropies are 0 when no code is active, as in :216-221."""
        cs = self.ema_cluster_size
        if not cs.is_cuda:
            raise RuntimeError("codebook_stats_device needs the module on the GPU")
        lib = load()
        out = torch.empty(5, dtype=torch.float64, device=cs.device)
        check(lib.vq_cluster_stats(ops._ptr(cs), ops._ptr(self.usage_count.to(cs.device)), self.num_embeddings,
                                   ops._ptr(out), ops._stream(cs.device)), "vq_cluster_stats")
        active = out[0] > 0
        zero = torch.zeros((), dtype=torch.float64, device=cs.device)
        return torch.stack([out[0], out[0] / self.num_embeddings, torch.where(active, out[2], zero),
                            torch.where(active, out[3], zero), out[4]])

    def last_step_counters(self) -> dict:
        """Counters of the most recent forward (one small device-to-host copy, i.e. a sync): rows that
        were not finite and went through the exact kernel -- the reference neither masks nor
        reports them (a NaN latent wins the argmin by torch's rule and is scattered into ema_w);
        this module keeps that behaviour and makes it visible -- for a temperature other than 1, the
        finite rows that had a code close above the minimum and were redone with every distance
        divided, and, for the filtered search, the rows its first and second pass left open."""
        if self._ws is None or "control" not in self._ws._bufs:
            return {"nonfinite_rows": 0, "codebook_not_finite": False, "filter_open_rows": 0, "filter_hard_rows": 0,
                    "temperature_close_rows": 0}
        # a finished step moves its counters from words 0..7 to words 8..15 and zeroes the former for the next step
        words = self._ws._bufs["control"][:64].view(torch.int32)
        c = (words[8:16] if self._ws.clean else words[:8]).cpu().tolist()
        return {"nonfinite_rows": int(c[0]), "codebook_not_finite": bool(c[1]), "filter_open_rows": int(c[4]),
                "filter_hard_rows": int(c[5]), "temperature_close_rows": int(c[6])}

    # ------------------------------------------------------------------ maintenance
    # These run a few times per training run on K-sized tensors.  They keep the reference's
    # semantics; the per-code Python loops become batched tensor ops, so the random draws come
    # from the same distributions but not the same generator sequence.
    def expand_codebook(self, new_size):
        """Grow the codebook (reference :110-163): new rows copy a random top-100-usage code plus
        N(0, 0.2^2) noise; EMA buffers are zero-extended."""
        if new_size <= self.num_embeddings:
            return False
        old_weight = self.embedding.weight.data
        old_size = self.num_embeddings
        device = old_weight.device
        self.num_embeddings = new_size
        new_embedding = nn.Embedding(new_size, self.embedding_dim).to(device)
        new_embedding.weight.data[:old_size] = old_weight
        if old_size > 0:
            usage = self.usage_count[:old_size].to(device)
            _, top = torch.topk(usage, min(100, old_size))
        self.embedding = new_embedding
        for name in ("ema_cluster_size", "ema_w", "usage_count"):
            old = getattr(self, name)
            new = torch.zeros((new_size,) + tuple(old.shape[1:]), device=old.device)
            new[:old_size] = old
            self.register_buffer(name, new)
        if old_size > 0:
            # new rows: a random top-usage code + N(0, 0.2^2); their EMA statistics stay zero (scale 0)
            self._reseed(torch.arange(old_size, new_size, device=device), top, 0.2, 0.0)
        print(f"Codebook expanded from {old_size} to {new_size}")
        return True

    def _reseed(self, targets: torch.Tensor, donors: torch.Tensor, noise: float, scale: float) -> None:
        """One kernel pair for all targets (Philox noise, seeded from torch's CPU generator)."""
        w = self.embedding.weight.data
        if not w.is_cuda:
            # state still on the host (module built or loaded before .to(device)): K-sized tensor ops
            pick = donors[torch.randint(0, len(donors), (len(targets),))]
            w[targets] = w[pick] + torch.randn(len(targets), self.embedding_dim) * noise
            self.ema_cluster_size[targets] = self.ema_cluster_size[pick] * scale
            self.ema_w[targets] = self.ema_w[pick] * scale
            self.usage_count[targets] = 0
            return
        self._follow(w.device)
        ops.reseed_codes(w, self.ema_w, self.ema_cluster_size, self.usage_count, targets, donors, noise, scale,
                         self._workspace(w.device))

    def reset_dead_codes(self, threshold=2, current_epoch=None):
        """Re-seed codes whose EMA cluster size is below ``threshold`` (reference :165-203)."""
        if not self.training or current_epoch is None:
            return 0
        if current_epoch - self.last_reset_epoch.item() < 3:
            return 0
        with torch.no_grad():
            dead = torch.where(self.ema_cluster_size < threshold)[0]
            n_dead = len(dead)
            if n_dead > 0:
                _, top = torch.topk(self.ema_cluster_size, min(100, self.num_embeddings - n_dead))
                if len(top) > 0:
                    self._reseed(dead, top, 0.2, 0.1)
                self.last_reset_epoch.fill_(current_epoch)
                print(f"Reset {n_dead} dead codes at epoch {current_epoch}")
            return n_dead

    def reset_low_usage_codes(self, percentage=0.1, current_epoch=None):
        """Re-seed the ``percentage`` least-used codes (reference :234-281)."""
        if not self.training or current_epoch is None:
            return 0
        if current_epoch - self.last_low_usage_reset_epoch.item() < 5:
            return 0
        with torch.no_grad():
            n_reset = max(1, int(self.num_embeddings * percentage))
            _, low = torch.topk(self.ema_cluster_size, n_reset, largest=False)
            if len(low) > 0:
                _, top = torch.topk(self.ema_cluster_size, min(100, self.num_embeddings - n_reset))
                if len(top) > 0:
                    self._reseed(low, top, 0.4, 0.2)
                self.last_low_usage_reset_epoch.fill_(current_epoch)
                print(f"Reset {len(low)} low usage codes at epoch {current_epoch}")
            return len(low)
