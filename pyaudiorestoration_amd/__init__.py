"""pyaudiorestoration_amd -- MI355X (gfx950) spectral-analysis + varispeed-resampling core.

Drop-in for the reference's util/ hot path: `fourier`, `resampling`, `wow_detection`, `filters`,
`correlation` keep the reference's function names and signatures; the work runs in hand-written
HIP kernels behind the C ABI of include/par_hip.h (libpar_hip.so, loaded with ctypes).
PyTorch-ROCm only owns device buffers and streams.  There is no CPU fallback: without the built
library or without a GPU every compute entry point raises.
"""
from . import _lib  # noqa: F401
from ._lib import ParError, ParUnsupported, ParIndexError, ParShapeError, ParEmptyBand  # noqa: F401

__all__ = ["fourier", "resampling", "wow_detection", "filters", "correlation", "pipeline", "io_ops",
           "ParError", "ParUnsupported", "ParIndexError", "ParShapeError", "ParEmptyBand"]
