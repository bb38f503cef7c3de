"""ctypes binding of libvqhip.so (include/vq_hip.h).

The library is the product: there is no fallback.  Loading fails loudly when the shared
object is missing, and every call checks the returned status.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
#: the library built in-tree; VQ_VAE_AMD_LIB selects another build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("VQ_VAE_AMD_LIB") or os.path.join(_PKG, "libvqhip.so")

LAYOUT_ND, LAYOUT_NCHW = 0, 1

#: every symbol include/vq_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vq_version": (c_int, []),
    "vq_status_string": (ctypes.c_char_p, [c_int]),
    "vq_init": (c_int, [c_int]),
    "vq_large_lds_kernels": (c_int, []),
    "vq_image_bytes": (c_size_t, [c_int, c_int]),
    "vq_norms_bytes": (c_size_t, [c_int]),
    "vq_search_blocks": (c_int64, [c_int64]),
    "vq_loss_partial_bytes": (c_size_t, [c_int64]),
    "vq_search_uses_persistent_kernel": (c_int, [c_int64, c_int, c_int, c_int, c_int64]),
    "vq_control_bytes": (c_size_t, []),
    "vq_search_split_bytes": (c_size_t, [c_int64, c_int]),
    "vq_scatter_slabs": (c_int, [c_int64, c_int, c_int]),
    "vq_control_reset": (c_int, [c_void_p, c_void_p]),
    "vq_codebook_prep": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vq_search": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_float,
                          c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p]),
    "vq_filter_image_bytes": (c_size_t, [c_int, c_int]),
    "vq_filter_work_bytes": (c_size_t, [c_int64]),
    "vq_filter_split": (c_int, [c_int64, c_int, c_void_p]),
    "vq_filter_prep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vq_filter_loss_partial_bytes": (c_size_t, [c_int64]),
    "vq_search_filtered": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_float, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "vq_search_exact": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_int, c_float, c_int64,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vq_gather_st": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "vq_code_histogram": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vq_scatter_slabs_for": (c_int, [c_int64, c_int, c_int, c_int, c_int64]),
    "vq_ema_scatter": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p,
                               c_void_p, c_int, c_void_p]),
    "vq_scatter_sorted_supported": (c_int, [c_int, c_int]),
    "vq_scatter_sorted_work_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vq_ema_scatter_sorted": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "vq_reduce_slabs": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vq_step_scalars": (c_int, [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "vq_ema_cluster_update": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                      c_void_p, c_void_p]),
    "vq_ema_codebook_update": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "vq_step_finish": (c_int, [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_int, c_float, c_float, c_float,
                               c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vq_ema_codebook_update_end": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "vq_quantize_step": (c_int, [c_void_p, c_void_p]),
    "vq_packed_stats_floats": (c_size_t, [c_int, c_int, c_int]),
    "vq_pack_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "vq_step_finish_reduced": (c_int, [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_int, c_float, c_float,
                                       c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "vq_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "vq_gather_project": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    "vq_backward_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_int,
                                   c_void_p, c_float, c_void_p, c_void_p]),
    "vq_conv1x1_image_bytes": (c_size_t, [c_int, c_int]),
    "vq_conv1x1_prep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vq_conv1x1": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "vq_conv1x1_wgrad_supported": (c_int, [c_int, c_int64, c_int, c_int, c_int64]),
    "vq_conv1x1_wgrad_work_bytes": (c_int64, [c_int64, c_int, c_int]),
    "vq_conv1x1_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "vq_reseed_stage_bytes": (c_size_t, [c_int, c_int]),
    "vq_reseed_codes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                c_float, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "vq_onehot": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vq_cluster_stats": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "vq_index_stats": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "vq_pairwise_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "vq_pairwise_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vq_pairwise_stats_static": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "vq_pack_dist_index": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vq_packed_min": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "vq_unpack_dist_index": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
}


class StepArgs(ctypes.Structure):
    """vq_step_args of include/vq_hip.h, field for field."""
    _fields_ = [("z", c_void_p), ("layout", c_int), ("N", c_int64), ("D", c_int), ("HW", c_int64),
                ("E", c_void_p), ("ema_w", c_void_p), ("cluster_size", c_void_p), ("usage", c_void_p), ("K", c_int),
                ("beta", c_float), ("decay", c_float), ("one_minus_decay", c_float), ("eps", c_float),
                ("k_eps", c_float), ("temperature", c_float),
                ("training", c_int),
                ("idx64", c_void_p), ("q_out", c_void_p), ("loss_out", c_void_p), ("perplexity_out", c_void_p),
                ("dmin", c_void_p),
                ("image", c_void_p), ("e2", c_void_p), ("control", c_void_p), ("idx32", c_void_p), ("nf_rows", c_void_p),
                ("loss_partial", c_void_p), ("split_ws", c_void_p),
                ("counts", c_void_p), ("counts_next", c_void_p), ("slabs", c_void_p), ("nslab", c_int),
                ("sort_work", c_void_p),
                ("control_clean", c_int)]


class VQLibraryError(RuntimeError):
    """A libvqhip.so entry point returned a non-zero status."""


_lib = None


def load() -> ctypes.CDLL:
    """Load libvqhip.so and attach the signatures.  Raises when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m vq_vae_amd.build` (hipcc, gfx950). "
            "The quantizer has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so and the header disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().vq_status_string(status).decode()
        raise VQLibraryError(f"{what} failed: status {status} ({msg})")
