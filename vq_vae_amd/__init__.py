"""MI355X-native vector quantizer: drop-in for the reference's vqvae.ema_vector_quantizer."""
from .ema_vector_quantizer import EMAVectorQuantizer, LazyOneHot  # noqa: F401
from .boundary import QuantizerBoundary  # noqa: F401

__all__ = ["EMAVectorQuantizer", "LazyOneHot", "QuantizerBoundary"]
__version__ = "0.1.0"
