"""Seeded input generators shared by the golden-vector script and the tests.

NumPy PCG64 streams, so the values do not depend on the torch build of the machine
that regenerates them.
"""
import numpy as np


def gauss(seed: int, shape) -> np.ndarray:
    """float32 N(0,1) samples from PCG64(seed)."""
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape, dtype=np.float32)


def ref_init_codebook(seed: int, K: int, D: int) -> np.ndarray:
    """U(-1/K, 1/K), the shape of the reference's codebook initialisation
    (vqvae/ema_vector_quantizer.py:22)."""
    u = np.random.Generator(np.random.PCG64(seed)).random((K, D), dtype=np.float32)
    return ((u * np.float32(2.0) - np.float32(1.0)) / np.float32(K)).astype(np.float32)
