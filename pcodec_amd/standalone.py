"""standalone.simple_* over the C ABI -- mirrors pco_python/src/standalone.rs:56-131 and
pco::standalone (standalone/simple.rs:22-152).  numpy arrays in, bytes out; the work happens on the GPU."""
import ctypes as C

import numpy as np

from . import _lib as G
from .config import ChunkConfig, Progress

_FILE_DTYPES = {1: np.uint32, 2: np.uint64, 3: np.int32, 4: np.int64, 5: np.float32, 6: np.float64,
                7: np.uint16, 8: np.int16, 9: np.float16, 10: np.uint8, 11: np.int8}


_TYPE_NAMES = {1: "U32", 2: "U64", 3: "I32", 4: "I64", 5: "F32", 6: "F64", 7: "U16", 8: "I16", 9: "F16", 10: "U8", 11: "I8"}


def _dtype_byte(arr):
    try:
        return G.DTYPE_BYTE[arr.dtype.name]
    except KeyError:
        raise TypeError(f"unsupported data type: {arr.dtype}")


def simple_compress(nums, config=None):
    """standalone::simple_compress (standalone/simple.rs:58-91): returns the .pco file as bytes."""
    nums = np.asarray(nums)
    if nums.ndim != 1:   # pco_python/src/utils.rs: the binding takes 1D contiguous arrays and says so (test_standalone.py:185-199)
        raise TypeError(f"{nums.ndim}D {nums.dtype} numpy array could not be cast to 1D")
    if not nums.flags["C_CONTIGUOUS"]:
        raise TypeError("nums is not contiguous")
    config = config or ChunkConfig()
    cfg = config.to_c()
    L = G.lib()
    dt = _dtype_byte(nums)
    n = C.c_size_t(0)
    exact = config.paging_spec.exact
    if exact is not None:   # PagingSpec::Exact: one chunk per entry (standalone/simple.rs:32-45)
        cap = L.pco_gfx_guarantee_file_size(0, dt, 0) + sum(L.pco_gfx_guarantee_chunk_size(max(int(s), 1), dt) for s in exact) + 64
        dst = np.empty(cap, np.uint8)
        sizes = (C.c_size_t * max(len(exact), 1))(*exact)
        G.check(L.pco_gfx_simple_compress_into_exact(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(dt), C.byref(cfg), C.c_int(0), sizes,
                                                     C.c_size_t(len(exact)), dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n)))
        return dst[: n.value].tobytes()
    cap = L.pco_gfx_guarantee_file_size(nums.size, dt, cfg.max_page_n) + 64
    dst = np.empty(cap, np.uint8)
    G.check(L.pco_gfx_simple_compress_into_ex(nums.ctypes.data_as(C.c_void_p), nums.size, dt, C.byref(cfg), 0,
                                              dst.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
    return dst[: n.value].tobytes()


def _peek_dtype_and_n_hint(data):
    """Enough of FileDecompressor::new + peek_number_type_or_termination (standalone/decompressor.rs:85-188)
    to size the output array: returns (dtype byte or 0 for an empty file, n_hint)."""
    b = bytes(data[:32])
    short = RuntimeError("pco error: pco InsufficientData error: the file ends inside its header")   # (bit_reader.rs: what the reference says of any truncated read)
    if len(b) < 4:
        raise short
    if b[:4] != b"pco!":
        raise RuntimeError("pco error: pco Corruption error: magic header does not match")
    if len(b) < 5:
        raise short
    ver = b[4]
    if ver < 2:
        return (b[5] if len(b) > 5 else 0), 0  # wrapped version byte follows; dtype byte after it
    pos = 5
    uniform = 0
    if ver >= 3:
        if pos >= len(b):
            raise short
        uniform = b[pos]; pos += 1
    bits = int.from_bytes(b[pos:pos + 10], "little")
    power = 1 + (bits & 63)
    n_hint = (bits >> 6) & ((1 << power) - 1)
    pos += (6 + power + 7) // 8
    if pos >= len(b):
        raise short
    major = b[pos]; pos += 1
    if major >= 4:
        pos += 1
        if pos > len(b):
            raise short
    if pos >= len(data):   # not even the terminator byte (standalone/decompressor.rs:190-200)
        raise short
    first = b[pos] if pos < len(b) else 0
    if uniform and first and first != uniform:   # standalone/decompressor.rs:200-210, with the reference's words
        raise RuntimeError(f"pco error: pco Corruption error: chunk's number type of {first} does not match file's uniform number type of "
                           f"{_TYPE_NAMES.get(uniform, uniform)}")
    return (uniform or first), n_hint


def simple_decompress(data):
    """standalone::simple_decompress (standalone/simple.rs:149-152): the dtype is read from the file."""
    data = bytes(data)
    dt, n_hint = _peek_dtype_and_n_hint(data)
    if dt == 0:
        return None  # empty file: pco_python returns None (standalone.rs:110-131)
    if dt not in _FILE_DTYPES:
        raise RuntimeError(f"unknown number type byte: {dt}")
    np_dtype = _FILE_DTYPES[dt]
    # n_hint is untrusted (standalone/decompressor.rs:265-277 caps its preallocation at DEFAULT_MAX_PREALLOC_BYTES): never size
    # the first attempt beyond 2^27 bytes or beyond what the file could plausibly hold (an all-constant chunk of 2^24 numbers is
    # ~40 bytes, so "plausible" is per chunk preamble, not per byte); the retry loop below grows on demand
    itemsize = np.dtype(np_dtype).itemsize
    plausible = max(1, len(data) // 8) * (1 << 24)
    cap = max(1, min(int(n_hint), (1 << 27) // itemsize, plausible))
    L = G.lib()
    buf = np.frombuffer(data, np.uint8)
    while True:
        out = np.empty(cap, np_dtype)
        n = C.c_size_t(0)
        code = L.pco_standalone_simple_decompress_into(buf.ctypes.data_as(C.c_void_p), len(buf), dt,
                                                       out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if code != G.PcoSuccess and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT and cap < (1 << 34):
            # n_hint is only a hint (standalone/decompressor.rs:265-277) -- but an honest writer puts the exact count there: a file that
            # outgrew the defensive first allocation is given what its header says next (one more pass, not up to four), else four times the room
            cap = min(int(n_hint), plausible) if cap < int(n_hint) <= plausible else cap * 4
            continue
        G.check(code)
        return out[: n.value].copy()


def simple_decompress_into(data, dst):
    """standalone::simple_decompress_into (standalone/simple.rs:100-143): fills `dst`, returns Progress."""
    data = bytes(data)
    dst_arr = np.asarray(dst)
    if not dst_arr.flags["C_CONTIGUOUS"] or dst_arr.ndim != 1:
        raise TypeError("dst must be a contiguous 1D array")
    full = simple_decompress(data)
    if full is None:
        return Progress(0, True)
    if full.dtype != dst_arr.dtype:
        raise RuntimeError(f"requested chunk decompression with {dst_arr.dtype} does not match chunk's number type of {full.dtype}")
    k = min(full.size, dst_arr.size)
    dst_arr[:k] = full[:k]
    return Progress(k, k == full.size)
