"""The pcodec-style Python surface (pcodec_amd.{standalone,wrapped}) exercised the way the reference's own Python tests do it
(pco_python/test/test_standalone.py, test_wrapped.py), on the GPU, with the oracle as the byte-level checker."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))

pytestmark = pytest.mark.gpu

ALL_DTYPES = ("f2", "f4", "f8", "i2", "i4", "i8", "u2", "u4", "u8")


def cfg_for(P, dtype, **kw):
    """The default ChunkConfig (every dtype, f16 included, goes through ModeSpec::Auto)."""
    return P.ChunkConfig(**kw)


@pytest.fixture(scope="module")
def P():
    import pcodec_amd as P
    from pcodec_amd import _lib as G
    assert G.lib().pco_gfx_device_count() >= 1, "these tests need an MI355X; the product has no CPU path"
    return P


@pytest.mark.parametrize("length", (0, 900))
@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_round_trip_decompress_into(P, length, dtype):   # test_standalone.py:24-36
    rng = np.random.default_rng(12345)
    data = rng.uniform(0, 1000, size=length).astype(dtype)
    compressed = P.standalone.simple_compress(data, cfg_for(P, dtype))
    out = np.empty_like(data)
    progress = P.standalone.simple_decompress_into(compressed, out)
    np.testing.assert_array_equal(data, out)
    assert progress.n_processed == data.size and progress.finished


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_round_trip_simple_decompress_paged(P, dtype):   # test_standalone.py:39-47
    rng = np.random.default_rng(7)
    data = rng.uniform(0, 1000, size=900).astype(dtype)
    compressed = P.standalone.simple_compress(data, cfg_for(P, dtype, paging_spec=P.PagingSpec.equal_pages_up_to(300)))
    np.testing.assert_array_equal(data, P.standalone.simple_decompress(compressed))


def test_inexact_decompression_and_type_mismatch(P):   # test_standalone.py:50-78
    rng = np.random.default_rng(3)
    data = rng.uniform(size=300)
    compressed = P.standalone.simple_compress(data, P.ChunkConfig())
    out = np.zeros(3)
    progress = P.standalone.simple_decompress_into(compressed, out)
    np.testing.assert_array_equal(out, data[:3])
    assert progress.n_processed == 3 and not progress.finished
    out = np.zeros(600)
    progress = P.standalone.simple_decompress_into(compressed, out)
    np.testing.assert_array_equal(out[:300], data)
    np.testing.assert_array_equal(out[300:], np.zeros(300))
    assert progress.n_processed == 300 and progress.finished
    with pytest.raises(RuntimeError, match="does not match chunk's number type"):
        P.standalone.simple_decompress_into(P.standalone.simple_compress(data.astype(np.float32), P.ChunkConfig()), np.zeros(100))


def test_compression_options(P):   # test_standalone.py:117-172 (Dict and Conv1 are outside the hot-path scope: refused loudly)
    from pcodec_amd import _lib as G
    rng = np.random.default_rng(5)
    data = rng.normal(size=100).astype(np.float32)
    default_size = len(P.standalone.simple_compress(data, P.ChunkConfig()))
    for delta_spec in (P.DeltaSpec.no_op(), P.DeltaSpec.try_consecutive(1), P.DeltaSpec.try_lookback()):
        compressed = P.standalone.simple_compress(data, P.ChunkConfig(compression_level=0, delta_spec=delta_spec, mode_spec=P.ModeSpec.classic(),
                                                                      paging_spec=P.PagingSpec.equal_pages_up_to(77)))
        np.testing.assert_array_equal(data, P.standalone.simple_decompress(compressed))
        assert len(compressed) >= default_size
    ints = (rng.normal(size=100) * 1000).astype(np.int32)
    for mode_spec in (P.ModeSpec.auto(), P.ModeSpec.classic(), P.ModeSpec.try_int_mult(10)):
        np.testing.assert_array_equal(ints, P.standalone.simple_decompress(P.standalone.simple_compress(ints, P.ChunkConfig(mode_spec=mode_spec))))
    floats = (rng.normal(size=100) * 1000).astype(np.int32) * np.pi
    for mode_spec in (P.ModeSpec.auto(), P.ModeSpec.classic(), P.ModeSpec.try_float_mult(10.0), P.ModeSpec.try_float_quant(4)):
        np.testing.assert_array_equal(floats, P.standalone.simple_decompress(P.standalone.simple_compress(floats, P.ChunkConfig(mode_spec=mode_spec))))
    for cfg in (P.ChunkConfig(mode_spec=P.ModeSpec.try_dict()), P.ChunkConfig(delta_spec=P.DeltaSpec.try_conv1(1))):
        with pytest.raises(G.PcoGfxError) as ei:
            P.standalone.simple_compress(ints, cfg)
        assert ei.value.status == G.ST_UNSUPPORTED


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_wrapped_compress(P, dtype):   # test_wrapped.py:11-52, with EqualPagesUpTo paging (6 + 4 numbers -> 2 pages of 5)
    import oracle_lib as O
    from pcodec_amd.wrapped import FileCompressor, FileDecompressor
    rng = np.random.default_rng(12345)
    data = rng.uniform(0, 1000, size=[10]).astype(dtype)
    pco_number_type = dtype[0].upper() + str(int(dtype[1]) * 8)
    fc = FileCompressor()
    header = fc.write_header()
    cc = fc.chunk_compressor(data, cfg_for(P, dtype, paging_spec=P.PagingSpec.equal_pages_up_to(6)))
    assert cc.n_per_page() == [5, 5]
    chunk_meta = cc.write_meta()
    page0, page1 = cc.write_page(0), cc.write_page(1)
    with pytest.raises(RuntimeError, match="page idx exceeds num pages"):
        cc.write_page(2)
    want_meta, want_pages, want_ns = O.wrapped_compress(data, O.make_config(max_page_n=6))
    assert (chunk_meta, [page0, page1], [5, 5]) == (want_meta, want_pages, want_ns)
    fd, n_bytes_read = FileDecompressor.new(header)
    assert n_bytes_read == len(header)
    _, n_bytes_read = FileDecompressor.new(header + b"foo")   # undershooting is fine
    assert n_bytes_read == len(header)
    cd, n_bytes_read = fd.chunk_decompressor(chunk_meta, pco_number_type)
    assert n_bytes_read == len(chunk_meta)
    dst1 = np.zeros(100).astype(dtype)
    _progress, n_bytes_read = cd.read_page_into(page1, 5, dst1)
    np.testing.assert_array_equal(dst1[5:], np.zeros(95))
    np.testing.assert_array_equal(dst1[:5], data[5:])
    assert n_bytes_read == len(page1)
    dst0 = np.zeros(5).astype(dtype)
    _progress, n_bytes_read = cd.read_page_into(page0, 5, dst0)
    np.testing.assert_array_equal(dst0, data[:5])
    assert n_bytes_read == len(page0)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_wrapped_compress_exact_page_sizes(P, dtype):   # test_wrapped.py:11-52 as written: PagingSpec.exact_page_sizes([6, 4])
    import oracle_lib as O
    from pcodec_amd.wrapped import FileCompressor, FileDecompressor
    rng = np.random.default_rng(12345)
    data = rng.uniform(0, 1000, size=[10]).astype(dtype)
    pco_number_type = dtype[0].upper() + str(int(dtype[1]) * 8)
    page_sizes = [6, 4]  # so there are 2 pages
    fc = FileCompressor()
    header = fc.write_header()
    cc = fc.chunk_compressor(data, P.ChunkConfig(paging_spec=P.PagingSpec.exact_page_sizes(page_sizes)))
    assert cc.n_per_page() == page_sizes
    chunk_meta = cc.write_meta()
    page0 = cc.write_page(0)
    page1 = cc.write_page(1)
    with pytest.raises(RuntimeError, match="page idx exceeds num pages"):
        cc.write_page(2)
    assert (chunk_meta, [page0, page1], page_sizes) == O.wrapped_compress(data, O.make_config(), exact_pages=page_sizes)
    fd, n_bytes_read = FileDecompressor.new(header)
    assert n_bytes_read == len(header)
    cd, n_bytes_read = fd.chunk_decompressor(chunk_meta, pco_number_type)
    assert n_bytes_read == len(chunk_meta)
    # page 1, which has elements 6-10
    dst1 = np.zeros(100).astype(dtype)
    _progress, n_bytes_read = cd.read_page_into(page1, 4, dst1)
    np.testing.assert_array_equal(dst1[4:], np.zeros(96))
    np.testing.assert_array_equal(dst1[:4], data[6:])
    assert n_bytes_read == len(page1)
    # page 0, which has elements 0-6
    dst0 = np.zeros(6).astype(dtype)
    _progress, n_bytes_read = cd.read_page_into(page0, 6, dst0)
    np.testing.assert_array_equal(dst0, data[:6])
    assert n_bytes_read == len(page0)
    # a wrong sum is the reference's InvalidArgument (chunk_config.rs:166-171)
    from pcodec_amd import _lib as G
    with pytest.raises(G.PcoGfxError, match="paging spec suggests 9 numbers but 10 were given"):
        fc.chunk_compressor(data, P.ChunkConfig(paging_spec=P.PagingSpec.exact_page_sizes([5, 4])))


def test_standalone_exact_page_sizes(P):   # standalone/simple.rs:32-45: the paging spec decides where the CHUNKS are cut
    import oracle_lib as O
    rng = np.random.default_rng(9)
    data = (rng.normal(size=3000) * 1000).astype(np.int32)
    sizes = [1000, 1, 1999]
    comp = P.standalone.simple_compress(data, P.ChunkConfig(paging_spec=P.PagingSpec.exact_page_sizes(sizes)))
    assert comp == O.simple_compress_exact(data, O.make_config(), sizes)
    np.testing.assert_array_equal(P.standalone.simple_decompress(comp), data)


def test_page_decompressor_partial_reads(P):   # page_decompressor.rs:193-221 through the Python mirror
    from pcodec_amd.wrapped import FileCompressor, FileDecompressor
    rng = np.random.default_rng(10)
    data = np.cumsum(rng.integers(-3, 9, 5000)).astype(np.int64)
    fc = FileCompressor()
    cc = fc.chunk_compressor(data, P.ChunkConfig(paging_spec=P.PagingSpec.exact_page_sizes([5000])))
    fd, _ = FileDecompressor.new(fc.write_header())
    cd, _ = fd.chunk_decompressor(cc.write_meta(), "I64")
    page = cc.write_page(0)
    pd = cd.page_decompressor(page, 5000)
    out = np.zeros(5120, np.int64); pos = 0
    for k in (256, 2048, 512):
        pr = pd.read(out[pos:pos + k])
        assert pr.n_processed == k and not pr.finished
        pos += k
    from pcodec_amd import _lib as G
    with pytest.raises(G.PcoGfxError, match="multiple of 256"):
        pd.read(np.zeros(100, np.int64))
    pr = pd.read(out[pos:pos + 2304])   # 2184 left: a multiple of 256 that is longer than the rest
    assert pr.n_processed == 5000 - pos and pr.finished
    np.testing.assert_array_equal(out[:5000], data)
    assert pd.n_bytes_read() == len(page)


def test_argument_errors(P):   # test_standalone.py:185-199
    rng = np.random.default_rng(1)
    with pytest.raises(TypeError, match="1D"):
        P.standalone.simple_compress(rng.normal(size=[10, 11]), P.ChunkConfig())
    with pytest.raises(TypeError, match="2D float64 numpy array could not be cast to 1D"):
        P.standalone.simple_compress(rng.normal(size=[10, 11]), P.ChunkConfig())
    with pytest.raises(TypeError, match="not contiguous"):   # test_standalone.py:201-203
        P.standalone.simple_compress(rng.normal(size=20)[::2], P.ChunkConfig())


def test_simple_decompress_errors_with_the_reference_words(P):   # test_standalone.py:81-114, on the reference's own asset
    path = os.path.join(HERE, "golden", "ref_assets", "v0_4_5_uniform_type.pco")
    compressed = bytearray(open(path, "rb").read())
    # byte 5 is the uniform number type, byte 8 the first chunk's number type
    with pytest.raises(RuntimeError, match="InsufficientData"):
        P.standalone.simple_decompress(bytes(compressed[:8]))
    compressed[8] = 99
    with pytest.raises(RuntimeError, match="chunk's number type of 99 does not match file's uniform number type of U32"):
        P.standalone.simple_decompress(bytes(compressed))
    compressed[8] = 0   # a file with a uniform type and no chunks: an empty array of that type
    out = P.standalone.simple_decompress(bytes(compressed))
    assert out.dtype == np.uint32 and out.size == 0
    compressed[5] = 0   # no uniform type, no chunk: None
    assert P.standalone.simple_decompress(bytes(compressed)) is None
    # every error of the library carries the reference's ErrorKind name in the reference's sentence (pco_python/src/utils.rs:78)
    good = bytearray(open(path, "rb").read())
    with pytest.raises(RuntimeError, match=r"pco error: pco (InsufficientData|Corruption) error"):
        P.standalone.simple_decompress(bytes(good[:-3]))


def test_decompress_without_n_hint(P):   # test_standalone.py:176-182: old files have no n_hint
    compressed = open(os.path.join(HERE, "golden", "ref_assets", "v0_0_0_classic.pco"), "rb").read()
    assert len(P.standalone.simple_decompress(compressed)) == 2000


def test_c_abi_round_trip_of_an_empty_array(P):   # standalone/simple.rs:22-48 with n = 0: header + terminator, and back (uniform type set, no chunk)
    import ctypes as C
    from pcodec_amd import _lib as G
    L = G.lib()
    dst = np.zeros(64, np.uint8); n = C.c_size_t(0)
    x = np.zeros(1, np.float64)
    assert L.pco_standalone_simple_compress_into(x.ctypes.data_as(C.c_void_p), 0, 6, None, dst.ctypes.data_as(C.c_void_p), 64, C.byref(n)) == G.PcoSuccess
    k = n.value; assert 0 < k <= 16 and dst[k - 1] == 0
    out = np.zeros(4, np.float64); m = C.c_size_t(99)
    assert L.pco_standalone_simple_decompress_into(dst.ctypes.data_as(C.c_void_p), k, 6, out.ctypes.data_as(C.c_void_p), 4, C.byref(m)) == G.PcoSuccess and m.value == 0
