"""wrapped.{FileCompressor, ChunkCompressor, FileDecompressor, ChunkDecompressor} over the C ABI -- mirrors
pco_python/src/wrapped/{compressor,decompressor}.rs (pcodec.wrapped) and pco::wrapped (wrapped/file_compressor.rs:54,
chunk_compressor.rs:442-705, file_decompressor.rs:24-60, page_decompressor.rs:193-246).  bytes / numpy arrays in and out; the
work happens on the GPU."""
import ctypes as C

import numpy as np

from . import _lib as G
from .config import ChunkConfig, Progress

_DTYPE_BY_NAME = {"U32": 1, "U64": 2, "I32": 3, "I64": 4, "F32": 5, "F64": 6, "U16": 7, "I16": 8, "F16": 9, "U8": 10, "I8": 11}
_NP_OF_BYTE = {1: np.uint32, 2: np.uint64, 3: np.int32, 4: np.int64, 5: np.float32, 6: np.float64, 7: np.uint16, 8: np.int16,
               9: np.float16, 10: np.uint8, 11: np.int8}


def _sizes(L):
    for f in ("pco_chunk_compressor_n_pages", "pco_chunk_compressor_page_n", "pco_chunk_compressor_meta_size_hint",
              "pco_chunk_compressor_page_size_hint", "pco_chunk_compressor_meta_size", "pco_chunk_compressor_page_size",
              "pco_page_decompressor_consumed"):
        getattr(L, f).restype = C.c_size_t


class FileCompressor:
    """wrapped::FileCompressor (wrapped/file_compressor.rs)."""

    def write_header(self):
        L = G.lib()
        buf = np.zeros(16, np.uint8)
        n = L.pco_wrapped_write_header(buf.ctypes.data_as(C.c_void_p), buf.size)
        return buf[:n].tobytes()

    def chunk_compressor(self, nums, config=None):
        nums = np.asarray(nums)
        if nums.ndim != 1:
            raise TypeError(f"{nums.ndim}D {nums.dtype} numpy array could not be cast to 1D")
        if not nums.flags["C_CONTIGUOUS"]:
            raise TypeError("nums is not contiguous")
        return ChunkCompressor(nums, config or ChunkConfig())


class ChunkCompressor:
    """wrapped::ChunkCompressor: the chunk is compressed at construction (chunk_compressor.rs:442), pages are handed out on demand."""

    def __init__(self, nums, config):
        L = G.lib(); _sizes(L)
        try:
            dt = G.DTYPE_BYTE[nums.dtype.name]
        except KeyError:
            raise TypeError(f"unsupported data type: {nums.dtype}")
        cfg = config.to_c(wrapped=True)
        self._h = C.c_void_p()
        exact = config.paging_spec.exact
        if exact is not None:   # PagingSpec::Exact (chunk_config.rs:124)
            sizes = (C.c_size_t * max(len(exact), 1))(*exact)
            G.check(L.pco_chunk_compressor_new_exact(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(dt), C.byref(cfg), sizes,
                                                     C.c_size_t(len(exact)), C.byref(self._h)))
        else:
            G.check(L.pco_chunk_compressor_new(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(dt), C.byref(cfg), C.byref(self._h)))
        self._L = L

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pco_chunk_compressor_free(self._h); self._h = None

    def n_per_page(self):  # chunk_compressor.rs:544
        n = self._L.pco_chunk_compressor_n_pages(self._h)
        return [int(self._L.pco_chunk_compressor_page_n(self._h, C.c_size_t(i))) for i in range(n)]

    def write_meta(self):  # :564
        cap = int(self._L.pco_chunk_compressor_meta_size(self._h))   # the exact length (the hint is the reference's estimate)
        buf = np.zeros(max(cap, 1), np.uint8); w = C.c_size_t(0)
        G.check(self._L.pco_chunk_compressor_write_meta(self._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(w)))
        return buf[: w.value].tobytes()

    def write_page(self, page_idx):  # :659
        if page_idx >= self._L.pco_chunk_compressor_n_pages(self._h):
            raise RuntimeError(f"page idx exceeds num pages ({page_idx} >= {self._L.pco_chunk_compressor_n_pages(self._h)})")
        # the exact stored length: page_size_hint is chunk-wide average bits x page_n x 1.2 and can undershoot a page that is less
        # compressible than the chunk's average (the reference only uses it to reserve() a growable Vec)
        cap = int(self._L.pco_chunk_compressor_page_size(self._h, C.c_size_t(page_idx)))
        buf = np.zeros(max(cap, 1), np.uint8); w = C.c_size_t(0)
        G.check(self._L.pco_chunk_compressor_write_page(self._h, C.c_size_t(page_idx), buf.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(w)))
        return buf[: w.value].tobytes()


class FileDecompressor:
    """wrapped::FileDecompressor (file_decompressor.rs:24-60)."""

    def __init__(self, major, minor):
        self.format_version = (major, minor)

    @staticmethod
    def new(src):
        L = G.lib()
        b = np.frombuffer(bytes(src), np.uint8)
        used = C.c_size_t(0); major = C.c_uint8(0); minor = C.c_uint8(0)
        G.check(L.pco_wrapped_read_header(b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.byref(used), C.byref(major), C.byref(minor)))
        return FileDecompressor(major.value, minor.value), int(used.value)

    def chunk_decompressor(self, src, dtype):
        if dtype not in _DTYPE_BY_NAME:
            raise RuntimeError(f"unknown number type: {dtype}")
        return ChunkDecompressor._new(bytes(src), _DTYPE_BY_NAME[dtype], self.format_version[0])


class ChunkDecompressor:
    """wrapped::ChunkDecompressor + PageDecompressor::read of whole pages (page_decompressor.rs:242)."""

    @staticmethod
    def _new(src, dt, major):
        L = G.lib()
        b = np.frombuffer(src, np.uint8)
        self = ChunkDecompressor.__new__(ChunkDecompressor)
        self._h = C.c_void_p(); self._L = L; self._dt = dt
        used = C.c_size_t(0)
        G.check(L.pco_chunk_decompressor_new(b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.c_ubyte(dt), C.c_uint8(major), C.byref(self._h), C.byref(used)))
        return self, int(used.value)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pco_chunk_decompressor_free(self._h); self._h = None

    def page_decompressor(self, src, page_n):
        """ChunkDecompressor::page_decompressor (chunk_decompressor.rs:74-80): a PageDecompressor over `src`."""
        return PageDecompressor(self, bytes(src), int(page_n))

    def read_page_into(self, src, page_n, dst):
        dst = np.asarray(dst)
        if dst.ndim != 1 or not dst.flags["C_CONTIGUOUS"]:
            raise TypeError("dst must be a contiguous 1D array")
        if dst.dtype != np.dtype(_NP_OF_BYTE[self._dt]):
            raise RuntimeError(f"requested chunk decompression with {dst.dtype} does not match chunk's number type of {np.dtype(_NP_OF_BYTE[self._dt])}")
        pd = self.page_decompressor(src, page_n)   # pco_python/src/wrapped/decompressor.rs:97-121: page_decompressor, read, into_src
        progress = pd.read(dst)
        return progress, pd.n_bytes_read()


class PageDecompressor:
    """wrapped::PageDecompressor (page_decompressor.rs:193-246): read() fills as much of dst as the page still holds; dst's length
    must be a multiple of 256 or at least the count of numbers remaining."""

    def __init__(self, cd, src, page_n):
        _sizes(cd._L)
        b = np.frombuffer(src, np.uint8)
        self._L = cd._L; self._dt = cd._dt; self._h = C.c_void_p()
        G.check(self._L.pco_page_decompressor_new(cd._h, b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.c_size_t(page_n), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pco_page_decompressor_free(self._h); self._h = None

    def read(self, dst):
        dst = np.asarray(dst)
        if dst.ndim != 1 or not dst.flags["C_CONTIGUOUS"]:
            raise TypeError("dst must be a contiguous 1D array")
        if dst.dtype != np.dtype(_NP_OF_BYTE[self._dt]):
            raise RuntimeError(f"requested chunk decompression with {dst.dtype} does not match chunk's number type of {np.dtype(_NP_OF_BYTE[self._dt])}")
        n_done = C.c_size_t(0); fin = C.c_int(0)
        G.check(self._L.pco_page_decompressor_read(self._h, dst.ctypes.data_as(C.c_void_p), C.c_size_t(dst.size), C.byref(n_done), C.byref(fin)))
        return Progress(int(n_done.value), bool(fin.value))

    def n_bytes_read(self):
        """into_src: how many bytes of `src` the page occupied."""
        return int(self._L.pco_page_decompressor_consumed(self._h))
