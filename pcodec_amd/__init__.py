"""pcodec_amd -- MI355X-native (gfx950 / HIP) implementation of pcodec's chunk encode / decode path.

The Python layer mirrors the reference's `pcodec` module (pco_python/src/{config,standalone,wrapped}.rs):
ChunkConfig / ModeSpec / DeltaSpec / PagingSpec / Progress, `standalone.simple_*` and `wrapped.{FileCompressor,
FileDecompressor}`, all routed through the C ABI of libpco_gfx.so (include/pco_gfx.h).  There is no CPU codec in this package.
"""
from .config import ChunkConfig, DeltaSpec, ModeSpec, PagingSpec, Progress  # noqa: F401
from . import standalone, wrapped  # noqa: F401

__all__ = ["ChunkConfig", "DeltaSpec", "ModeSpec", "PagingSpec", "Progress", "standalone", "wrapped"]
