"""-m gpu: PagingSpec::Exact (chunk_config.rs:124,162-180) and batch-granular PageDecompressor::read
(wrapped/page_decompressor.rs:193-246) through the C ABI, compared with the oracle's wrapped chunk byte for byte."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))

import oracle_lib as O  # noqa: E402
import gpu_util as U  # noqa: E402
from pcodec_amd import _lib as G  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = G.lib()
    assert lib.pco_gfx_device_count() >= 1, "these tests need an MI355X; the product has no CPU path"
    for f in ("pco_chunk_compressor_n_pages", "pco_chunk_compressor_page_n", "pco_chunk_compressor_meta_size", "pco_chunk_compressor_page_size",
              "pco_page_decompressor_consumed"):
        getattr(lib, f).restype = C.c_size_t
    return lib


def gpu_wrapped_exact(L, nums, cfg, sizes):
    """(meta, [pages], [page_n]) of pco_chunk_compressor_new_exact."""
    cc = C.c_void_p()
    arr = (C.c_size_t * max(len(sizes), 1))(*[int(s) for s in sizes])
    G.check(L.pco_chunk_compressor_new_exact(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(G.DTYPE_BYTE[nums.dtype.name]), C.byref(cfg), arr,
                                             C.c_size_t(len(sizes)), C.byref(cc)))
    try:
        n_pages = L.pco_chunk_compressor_n_pages(cc)
        w = C.c_size_t(0)
        buf = np.zeros(L.pco_chunk_compressor_meta_size(cc) + 8, np.uint8)
        G.check(L.pco_chunk_compressor_write_meta(cc, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w)))
        meta = bytes(buf[: w.value]); pages = []; page_ns = []
        for i in range(n_pages):
            exact_len = L.pco_chunk_compressor_page_size(cc, C.c_size_t(i))
            buf = np.zeros(exact_len + 8, np.uint8)
            G.check(L.pco_chunk_compressor_write_page(cc, C.c_size_t(i), buf.ctypes.data_as(C.c_void_p), C.c_size_t(exact_len), C.byref(w)))   # the exact size suffices
            assert w.value == exact_len
            pages.append(bytes(buf[: w.value])); page_ns.append(int(L.pco_chunk_compressor_page_n(cc, C.c_size_t(i))))
        return meta, pages, page_ns
    finally:
        L.pco_chunk_compressor_free(cc)


def decode_pages(L, meta, pages, page_ns, np_dtype):
    """Every page through pco_chunk_decompressor_read_page, in REVERSE order (pages are independent: tests/low_level.rs)."""
    dt = G.DTYPE_BYTE[np.dtype(np_dtype).name]
    cd = C.c_void_p(); used = C.c_size_t(0)
    mbuf = np.frombuffer(meta, np.uint8)
    G.check(L.pco_chunk_decompressor_new(mbuf.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), C.c_ubyte(dt), C.c_uint8(4), C.byref(cd), C.byref(used)))
    assert used.value == len(meta)
    out = [None] * len(pages)
    try:
        for i in reversed(range(len(pages))):
            pb = np.frombuffer(pages[i] + b"\x00" * 3, np.uint8)   # trailing bytes of the next page must not be touched
            dst = np.zeros(page_ns[i], np_dtype); npr = C.c_size_t(0); cons = C.c_size_t(0)
            G.check(L.pco_chunk_decompressor_read_page(cd, pb.ctypes.data_as(C.c_void_p), C.c_size_t(pb.size), C.c_size_t(page_ns[i]), dst.ctypes.data_as(C.c_void_p),
                                                       C.c_size_t(dst.size), C.byref(npr), C.byref(cons)))
            assert npr.value == page_ns[i] and cons.value == len(pages[i]), (i, npr.value, cons.value, len(pages[i]))
            out[i] = dst
    finally:
        L.pco_chunk_decompressor_free(cd)
    return np.concatenate(out)


def data_for(kind, n, rng):
    if kind == "i64walk":
        return (np.cumsum(rng.integers(-20, 90, n)) + (1 << 35)).astype(np.int64)
    if kind == "u32mult":
        return (rng.integers(0, 5000, n) * 77 + rng.integers(0, 2, n)).astype(np.uint32)
    if kind == "f64cents":
        return rng.integers(1000, 10000, n) / 100.0
    if kind == "f32quant":
        x = rng.normal(size=n).astype(np.float32)
        return (x.view(np.uint32) & np.uint32(0xFFFFF000)).view(np.float32)
    if kind == "i16season":
        base = rng.integers(-20000, 20000, 97)
        return (base[np.arange(n) % 97] + rng.integers(-2, 3, n)).astype(np.int16)
    if kind == "f16":
        return rng.uniform(0, 100, n).astype(np.float16)
    raise KeyError(kind)


EXACT_CASES = [
    # (data kind, config kwargs): every two-variable mode, each delta kind, Auto
    ("i64walk", dict(mode=1, delta=2, delta_order=1)),
    ("i64walk", dict(mode=1, delta=1)),
    ("i64walk", dict(mode=1, delta=3)),
    ("u32mult", dict(mode=4, mode_u64=77, delta=1)),
    ("u32mult", dict(mode=4, mode_u64=77, delta=2, delta_order=2)),
    ("u32mult", dict(mode=4, mode_u64=77, delta=3)),
    ("f64cents", dict(mode=2, mode_f64=0.01, delta=1)),
    ("f64cents", dict(mode=2, mode_f64=0.01, delta=2, delta_order=1)),
    ("f32quant", dict(mode=3, mode_u64=12, delta=1)),
    ("f32quant", dict(mode=3, mode_u64=12, delta=3)),
    ("i16season", dict(mode=1, delta=3)),
    ("f16", dict()),
    ("i64walk", dict()),
    ("f64cents", dict()),
]
PAGE_LISTS = [
    [1, 9999, 300, 257, 256, 255, 1, 5000, 4931],   # 1-number pages, around a batch boundary, sizes that are no multiple of 256
    [20000],                                           # one page
    [17, 19983],
    [6000, 1, 1, 1, 13997],
]


@pytest.mark.parametrize("case", range(len(EXACT_CASES)))
def test_exact_paging_matches_the_oracle(L, case):
    """pco_chunk_compressor_new_exact against the oracle's ChunkCompressor with PagingSpec::Exact: meta and every page byte-identical,
    every page decodes (in any order) to its slice.  Delta state restarts on every page (page_meta: delta moments / lookback state)."""
    kind, kw = EXACT_CASES[case]
    rng = np.random.default_rng(1000 + case)
    for sizes in PAGE_LISTS:
        n = sum(sizes)
        nums = data_for(kind, n, rng)
        kw8 = dict(kw, enable_8_bit=True)
        want_meta, want_pages, want_ns = O.wrapped_compress(nums, O.make_config(**kw8), exact_pages=sizes)
        meta, pages, page_ns = gpu_wrapped_exact(L, nums, G.make_config(**kw8), sizes)
        assert page_ns == list(sizes) == want_ns, (kind, kw, sizes)
        assert meta == want_meta, (kind, kw, sizes)
        for i, (a, b) in enumerate(zip(pages, want_pages)):
            assert a == b, (kind, kw, sizes, i, len(a), len(b))
        assert U.bits_equal(decode_pages(L, meta, pages, page_ns, nums.dtype), nums), (kind, kw, sizes)


def test_exact_paging_full_size_chunk(L):
    """One 2^18 + 77 chunk cut into uneven exact pages (a full 2^18-less-one page, a 1-number page, a short tail)."""
    rng = np.random.default_rng(5)
    sizes = [(1 << 17) + 3, 1, (1 << 17) - 4 + 77]
    nums = U.synth("c2", n=sum(sizes), seed=5)
    kw = dict(mode=1, delta=2, delta_order=1)
    want = O.wrapped_compress(nums, O.make_config(**kw), exact_pages=sizes)
    got = gpu_wrapped_exact(L, nums, G.make_config(**kw), sizes)
    assert got == want
    assert U.bits_equal(decode_pages(L, got[0], got[1], got[2], nums.dtype), nums)
    del rng


def test_exact_paging_argument_errors(L):
    """chunk_config.rs:162-180: the sizes must sum to n and none may be 0 -- InvalidArgument, like the oracle."""
    nums = np.arange(100, dtype=np.uint32)
    cfg = G.make_config(mode=1, delta=1)
    for sizes in ([50, 49], [50, 51], [100, 0], [0, 100], []):
        with pytest.raises(G.PcoGfxError) as ei:
            gpu_wrapped_exact(L, nums, cfg, sizes)
        assert ei.value.status == G.ST_INVALID_ARGUMENT, sizes
        with pytest.raises(O.OracleError) as oi:
            O.wrapped_compress(nums, O.make_config(mode=1, delta=1), exact_pages=sizes)
        assert oi.value.kind == O.ERR_INVALID_ARGUMENT, sizes


@pytest.mark.parametrize("kw", [dict(mode=1, delta=2, delta_order=1), dict(mode=1, delta=3), dict(mode=2, mode_f64=0.01, delta=1)])
def test_partial_page_reads(L, kw):
    """PageDecompressor::read (page_decompressor.rs:193-221): a 70 000-number page handed out in multiples of 256 of varying
    size; Progress after every call; a length that is neither a multiple of 256 nor >= the rest is InvalidArgument and
    changes nothing; into_src position (consumed) at the end."""
    rng = np.random.default_rng(70)
    n = 70000
    nums = data_for("f64cents" if kw.get("mode") == 2 else "i64walk", n + 3000, rng)
    sizes = [3000, n]
    meta, pages, page_ns = gpu_wrapped_exact(L, nums, G.make_config(**kw), sizes)
    want = nums[3000:]
    dt = G.DTYPE_BYTE[nums.dtype.name]
    cd = C.c_void_p(); used = C.c_size_t(0)
    mbuf = np.frombuffer(meta, np.uint8)
    G.check(L.pco_chunk_decompressor_new(mbuf.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), C.c_ubyte(dt), C.c_uint8(4), C.byref(cd), C.byref(used)))
    pb = np.frombuffer(pages[1] + b"\xff" * 5, np.uint8)
    pd = C.c_void_p()
    G.check(L.pco_page_decompressor_new(cd, pb.ctypes.data_as(C.c_void_p), C.c_size_t(pb.size), C.c_size_t(n), C.byref(pd)))
    try:
        got = []; pos = 0
        lens = [256, 1024, 256 * 7, 0, 256 * 100, 512, 256 * 33, 256]   # (a zero-length dst is a multiple of 256: nothing happens)
        npr = C.c_size_t(0); fin = C.c_int(0)
        for k in lens:
            dst = np.zeros(max(k, 1), nums.dtype)
            G.check(L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(k), C.byref(npr), C.byref(fin)))
            assert npr.value == k and fin.value == 0, (k, npr.value, fin.value)
            got.append(dst[:k].copy()); pos += k
        remaining = n - pos
        assert remaining > 300
        bad = np.zeros(300, nums.dtype)
        code = L.pco_page_decompressor_read(pd, bad.ctypes.data_as(C.c_void_p), C.c_size_t(300), C.byref(npr), C.byref(fin))
        assert code == G.PcoDecompressionError and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
        assert b"multiple of 256" in L.pco_gfx_last_error()
        # the failed call consumed nothing: go on in batches until fewer than 256 numbers are left, then take the rest with a
        # destination that is larger than what remains and no multiple of 256
        while n - pos > 256 * 3:
            dst = np.zeros(256 * 3, nums.dtype)
            G.check(L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(dst.size), C.byref(npr), C.byref(fin)))
            assert npr.value == 768 and fin.value == 0
            got.append(dst.copy()); pos += 768
        rest = n - pos
        dst = np.zeros(rest + 5, nums.dtype)
        G.check(L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(dst.size), C.byref(npr), C.byref(fin)))
        assert npr.value == rest and fin.value == 1
        assert not dst[rest:].view(np.uint8).any(), "read wrote past the numbers it reported"
        got.append(dst[:rest].copy())
        assert U.bits_equal(np.concatenate(got), want)
        # a finished page: any further read reports (0, finished)
        G.check(L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(7), C.byref(npr), C.byref(fin)))
        assert npr.value == 0 and fin.value == 1
        assert L.pco_page_decompressor_consumed(pd) == len(pages[1])
    finally:
        L.pco_page_decompressor_free(pd)
        L.pco_chunk_decompressor_free(cd)


def test_partial_read_exact_multiple_and_short_pages(L):
    """Pages of exactly 512 numbers (the last full batch finishes the page), of 1 number and of 255 numbers."""
    rng = np.random.default_rng(71)
    sizes = [512, 1, 255]
    nums = data_for("i64walk", sum(sizes), rng)
    meta, pages, page_ns = gpu_wrapped_exact(L, nums, G.make_config(mode=1, delta=2, delta_order=2), sizes)
    cd = C.c_void_p(); used = C.c_size_t(0)
    mbuf = np.frombuffer(meta, np.uint8)
    G.check(L.pco_chunk_decompressor_new(mbuf.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), C.c_ubyte(4), C.c_uint8(4), C.byref(cd), C.byref(used)))
    npr = C.c_size_t(0); fin = C.c_int(0); start = 0
    try:
        for i, pn in enumerate(sizes):
            pb = np.frombuffer(pages[i], np.uint8); pd = C.c_void_p()
            G.check(L.pco_page_decompressor_new(cd, pb.ctypes.data_as(C.c_void_p), C.c_size_t(pb.size), C.c_size_t(pn), C.byref(pd)))
            dst = np.zeros(256, np.int64); out = []
            while True:
                G.check(L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(256), C.byref(npr), C.byref(fin)))
                out.append(dst[: npr.value].copy())
                if fin.value:
                    break
                assert npr.value == 256
            assert np.array_equal(np.concatenate(out), nums[start:start + pn]), i
            assert L.pco_page_decompressor_consumed(pd) == len(pages[i])
            L.pco_page_decompressor_free(pd); start += pn
    finally:
        L.pco_chunk_decompressor_free(cd)


def test_standalone_exact_paging(L):
    """standalone::simple_compress under PagingSpec::Exact cuts one CHUNK per entry (standalone/simple.rs:32-45)."""
    rng = np.random.default_rng(72)
    L.pco_gfx_guarantee_chunk_size.restype = C.c_size_t
    for kind, kw, sizes in (("i64walk", dict(mode=1, delta=2, delta_order=1), [1, 300, 4000, 255, 1]), ("f64cents", dict(), [700, 9000]),
                            ("u32mult", dict(mode=4, mode_u64=77, delta=3), [5000, 5000, 123])):
        nums = data_for(kind, sum(sizes), rng)
        dt = G.DTYPE_BYTE[nums.dtype.name]
        for uniform in (0, 1):
            want = O.simple_compress_exact(nums, O.make_config(**kw), sizes, uniform_type=bool(uniform))
            cap = L.pco_gfx_guarantee_file_size(0, dt, 0) + sum(L.pco_gfx_guarantee_chunk_size(s, dt) for s in sizes)
            dst = np.zeros(cap, np.uint8); w = C.c_size_t(0)
            arr = (C.c_size_t * len(sizes))(*sizes)
            cfg = G.make_config(**kw)
            G.check(L.pco_gfx_simple_compress_into_exact(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(dt), C.byref(cfg), C.c_int(uniform), arr,
                                                         C.c_size_t(len(sizes)), dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(w)))
            got = bytes(dst[: w.value])
            assert got == want, (kind, kw, uniform)
            assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums)
    nums = np.arange(10, dtype=np.uint32); cfg = G.make_config(mode=1, delta=1)
    for sizes in ([5, 4], [10, 0]):
        arr = (C.c_size_t * len(sizes))(*sizes); dst = np.zeros(4096, np.uint8); w = C.c_size_t(0)
        code = L.pco_gfx_simple_compress_into_exact(nums.ctypes.data_as(C.c_void_p), C.c_size_t(10), C.c_ubyte(1), C.byref(cfg), C.c_int(0), arr, C.c_size_t(2),
                                                    dst.ctypes.data_as(C.c_void_p), C.c_size_t(4096), C.byref(w))
        assert code == G.PcoCompressionError and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT


def test_a_page_that_fails_in_batch_k_hands_out_the_batches_before_k(L):
    """wrapped/page_decompressor.rs:115-221: the reference decodes a page batch by batch, so a truncated page yields its intact batches and
    fails on the call that reaches the first batch it cannot finish; a page whose own metadata is cut fails in PageDecompressor::new.  The GPU
    decodes the page when the handle is made and reproduces that timing: same number of good numbers, same numbers, same error kind as the
    oracle's batch-by-batch decoder, for reads of one batch, of several, and of the whole rest."""
    rng = np.random.default_rng(21)
    cases = [(U.synth("c2", 70000), dict(mode=1, delta=2, delta_order=1)),
             ((rng.integers(1000, 10000, 40000) / 100.0), dict(mode=2, mode_f64=0.01, delta=1)),
             (rng.integers(0, 1 << 32, 5000, dtype=np.uint64).astype(np.uint32), dict(mode=1, delta=1))]
    for nums, kw in cases:
        meta, pages, page_ns = gpu_wrapped_exact(L, nums, G.make_config(**kw), [nums.size])
        page = pages[0]; n = page_ns[0]
        dt = G.DTYPE_BYTE[nums.dtype.name]
        mbuf = np.frombuffer(meta, np.uint8)
        cd = C.c_void_p(); used = C.c_size_t(0)
        G.check(L.pco_chunk_decompressor_new(mbuf.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), C.c_ubyte(dt), C.c_uint8(4), C.byref(cd), C.byref(used)))
        try:
            for cut in sorted({3, 9, len(page) // 7, len(page) // 3, len(page) // 2, (len(page) * 9) // 10, len(page) - 1}):
                want, err, in_meta = O.wrapped_page_prefix(meta, page[:cut], nums.dtype, n)
                assert err == 2, (cut, err)   # InsufficientData
                pbuf = np.frombuffer(page[:cut] if cut else b"\0", np.uint8)
                for step in (256, 1024, n):   # one batch a call, four, everything that is left
                    pd = C.c_void_p()
                    code = L.pco_page_decompressor_new(cd, pbuf.ctypes.data_as(C.c_void_p), C.c_size_t(cut), C.c_size_t(n), C.byref(pd))
                    if in_meta:
                        assert code != 0 and L.pco_gfx_last_status() == G.ST_INSUFFICIENT_DATA, (cut, step)
                        continue
                    assert code == 0, (cut, step, L.pco_gfx_last_status())
                    try:
                        got = []
                        while True:
                            dst = np.zeros(step, nums.dtype); k = C.c_size_t(0); fin = C.c_int(0)
                            code = L.pco_page_decompressor_read(pd, dst.ctypes.data_as(C.c_void_p), C.c_size_t(step), C.byref(k), C.byref(fin))
                            if code != 0:
                                assert L.pco_gfx_last_status() == G.ST_INSUFFICIENT_DATA
                                break
                            got.append(dst[: k.value]); assert not fin.value
                        got = np.concatenate(got) if got else np.zeros(0, nums.dtype)
                        # whole calls only: the call that reaches the bad batch fails, whatever it had decoded in front of it
                        assert got.size == (want.size // step) * step, (cut, step, got.size, want.size)
                        assert U.bits_equal(got, want[: got.size])
                    finally:
                        L.pco_page_decompressor_free(pd)
        finally:
            L.pco_chunk_decompressor_free(cd)


def test_many_pages_of_periodic_data_repeatedly():
    """Twenty pages of 37-periodic bytes (51 bins, ans_size_log 9: tANS fields of 3..7 bits) through the standalone entry point, over and over:
    the case in which a build of enc_walkp_kernel (round 6) garbled the tANS section of one batch in a few -- nine runs in ten differed -- while
    every other test passed.  Bytes identical to the oracle's every time; every page's batches through walk + pack."""
    rng = np.random.default_rng(77)
    base = rng.integers(0, 250, 37)
    for dt, n, page in ((np.uint8, 139680, 6984), (np.uint16, 69840, 3492), (np.uint8, 70000, 6984)):
        nums = ((base[np.arange(n) % 37] + rng.integers(0, 3, n)) % 256).astype(dt)
        kw = dict(level=7, mode=1, delta=1, max_page_n=page, enable_8_bit=True)
        want = O.simple_compress(nums, O.make_config(**kw))
        for rep in range(12):
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            assert got == want, (np.dtype(dt).name, n, page, rep)
