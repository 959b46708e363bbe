"""CPU-side checks of the drop-in boundary: libpco_gfx.so loads, exports every symbol that
include/pco_gfx.h declares, keeps the reference's struct layouts, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pcodec_amd import _lib as G
from pcodec_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if B.needs_build():
        B.build()
    return G.lib()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pco_gfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(pco_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/pco_gfx.h but not exported: {missing}"
    for ref in ("pco_standalone_guarantee_file_size", "pco_standalone_simple_compress_into", "pco_standalone_simple_decompress_into"):
        assert ref in names  # the reference's own three entry points (pco_c/include/cpcodec_generated.h)


def test_struct_layouts_match_header():
    assert C.sizeof(G.PcoChunkConfig) == 16          # unsigned + size_t (pco_c/src/lib.rs:21-32)
    assert C.sizeof(G.PcoChunkConfigEx) == 48
    assert C.sizeof(G.EncodeTask) == 40 and C.sizeof(G.DecodeTask) == 40 and C.sizeof(G.TaskResult) == 24


def test_size_guarantees_match_the_oracle(lib, oracle):
    for dt in range(1, 12):
        for n in (0, 1, 255, 1 << 18, (1 << 18) + 1, 1000003):
            assert lib.pco_standalone_guarantee_file_size(n, dt) == oracle.file_size_bound(n, dt)
            assert lib.pco_gfx_guarantee_file_size(n, dt, 1000) == oracle.file_size_bound(n, dt, 1000)
    assert lib.pco_standalone_guarantee_file_size(10, 0) == 0 and lib.pco_standalone_guarantee_file_size(10, 12) == 0


def test_framing_matches_the_oracle(lib, oracle):
    # standalone header / footer bytes (standalone/compressor.rs:85-105,157-162)
    buf = np.zeros(64, np.uint8)
    for n in (0, 1, 5, 129, 1000, 1 << 18, (1 << 24) + 7):
        for uniform in (0, 1):
            nums = np.arange(min(n, 20), dtype=np.uint32)
            k = lib.pco_gfx_write_standalone_header(buf.ctypes.data_as(C.c_void_p), 64, n, 1 if uniform else 0)
            if n <= 20:
                want = oracle.simple_compress(nums, oracle.make_config(mode=1, delta=1), uniform_type=bool(uniform))
                assert bytes(buf[:k]) == want[:k]
            assert bytes(buf[:5]) == b"pco!\x03" and buf[5] == uniform and bytes(buf[k - 2:k]) == b"\x04\x01"
    assert lib.pco_gfx_write_standalone_footer(buf.ctypes.data_as(C.c_void_p), 64) == 1 and buf[0] == 0
    assert lib.pco_wrapped_write_header(buf.ctypes.data_as(C.c_void_p), 64) == 2 and bytes(buf[:2]) == b"\x04\x01"
    assert lib.pco_gfx_write_standalone_header(buf.ctypes.data_as(C.c_void_p), 3, 5, 0) == 0  # too small


def test_no_cpu_fallback(lib):
    """Without a HIP device every compute entry point must fail loudly (never route to a CPU codec)."""
    if lib.pco_gfx_device_count() > 0:
        pytest.skip("a GPU is visible here")
    nums = np.arange(100, dtype=np.uint32); dst = np.zeros(4096, np.uint8); n = C.c_size_t(0)
    code = lib.pco_standalone_simple_compress_into(nums.ctypes.data_as(C.c_void_p), 100, 1, None, dst.ctypes.data_as(C.c_void_p), 4096, C.byref(n))
    assert code == G.PcoCompressionError and lib.pco_gfx_last_status() == G.ST_DEVICE_ERROR
    code = lib.pco_standalone_simple_decompress_into(dst.ctypes.data_as(C.c_void_p), 10, 1, nums.ctypes.data_as(C.c_void_p), 100, C.byref(n))
    assert code == G.PcoDecompressionError and lib.pco_gfx_last_status() == G.ST_DEVICE_ERROR
    assert b"no CPU fallback" in lib.pco_gfx_last_error()
    code = lib.pco_standalone_simple_compress_into(nums.ctypes.data_as(C.c_void_p), 100, 77, None, dst.ctypes.data_as(C.c_void_p), 4096, C.byref(n))
    assert code == G.PcoInvalidType


def test_product_does_not_touch_the_oracle():
    """The product sources must not include / link / import anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "pcodec_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".inc", ".py", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle[/_]|pco_oracle|oracle_lib", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    import subprocess
    out = subprocess.run(["ldd", G.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_python_mirror_surface():
    import pcodec_amd as P
    c = P.ChunkConfig(compression_level=5, mode_spec=P.ModeSpec.try_int_mult(8), delta_spec=P.DeltaSpec.try_consecutive(2),
                      paging_spec=P.PagingSpec.equal_pages_up_to(1000), enable_8_bit=True).to_c()
    assert (c.compression_level, c.mode_kind, c.mode_u64, c.delta_kind, c.delta_order, c.max_page_n, c.enable_8_bit) == (5, 4, 8, 2, 2, 1000, 1)
    for name in ("simple_compress", "simple_decompress", "simple_decompress_into"):
        assert callable(getattr(P.standalone, name))
    with pytest.raises(RuntimeError):
        P.standalone.simple_decompress(b"nope!....")


def test_python_surface_host_side(lib):
    """What of pcodec_amd.{standalone,wrapped} is host-only framing works without a GPU; what computes fails loudly."""
    import pcodec_amd as P
    from pcodec_amd.wrapped import FileCompressor, FileDecompressor
    header = FileCompressor().write_header()
    assert header == b"\x04\x01"                       # wrapped header = format version (file_compressor.rs:54)
    fd, used = FileDecompressor.new(header + b"tail")
    assert used == 2 and fd.format_version == (4, 1)
    with pytest.raises(G.PcoGfxError):
        FileDecompressor.new(b"\x09")                  # a major version from the future (format_version.rs)
    with pytest.raises(TypeError):
        FileCompressor().chunk_compressor(np.zeros((2, 2)), P.ChunkConfig())
    with pytest.raises(RuntimeError, match="unknown number type"):
        fd.chunk_decompressor(b"", "U128")
    assert P.ChunkConfig(paging_spec=P.PagingSpec.exact_page_sizes([6, 4])).paging_spec.exact == (6, 4)
    if lib.pco_gfx_device_count() == 0:
        with pytest.raises(G.PcoGfxError) as ei:
            P.standalone.simple_compress(np.arange(10, dtype=np.uint32), P.ChunkConfig())
        assert ei.value.status == G.ST_DEVICE_ERROR
        with pytest.raises(G.PcoGfxError) as ei:
            FileCompressor().chunk_compressor(np.arange(10, dtype=np.uint32), P.ChunkConfig())
        assert ei.value.status == G.ST_DEVICE_ERROR
