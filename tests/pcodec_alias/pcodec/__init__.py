"""TESTS-ONLY alias: makes `pcodec_amd` importable under the reference Python package's name, so that the reference's own
pco_python/test/*.py (which `from pcodec import ChunkConfig, ...; from pcodec.wrapped import FileCompressor, FileDecompressor`) can run
UNMODIFIED against libpco_gfx.so (tests/test_reference_clients.py puts this directory on PYTHONPATH for that run and for nothing else).
Not part of the product: a user of the product imports pcodec_amd."""
import sys

import pcodec_amd
from pcodec_amd import ChunkConfig, DeltaSpec, ModeSpec, PagingSpec, Progress, standalone, wrapped  # noqa: F401

sys.modules[__name__ + ".standalone"] = standalone
sys.modules[__name__ + ".wrapped"] = wrapped
__all__ = list(pcodec_amd.__all__)
