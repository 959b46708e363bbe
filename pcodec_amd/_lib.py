"""ctypes binding of libpco_gfx.so (the C ABI declared in include/pco_gfx.h).

The shared library is the product; this module only loads it.  There is no Python or CPU codec
behind it: if the library is missing, or no HIP device is visible, calls fail loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCO_GFX_LIB", os.path.join(HERE, "libpco_gfx.so"))  # override only for A/B experiments

PcoSuccess, PcoInvalidType, PcoCompressionError, PcoDecompressionError = range(4)
ST_OK, ST_CORRUPTION, ST_INSUFFICIENT_DATA, ST_INVALID_ARGUMENT, ST_UNSUPPORTED, ST_DEVICE_ERROR = range(6)
TASK_HAS_FILE_HEADER = 1

MODE_AUTO, MODE_CLASSIC, MODE_TRY_FLOAT_MULT, MODE_TRY_FLOAT_QUANT, MODE_TRY_INT_MULT, MODE_TRY_DICT = range(6)
DELTA_AUTO, DELTA_NOOP, DELTA_TRY_CONSECUTIVE, DELTA_TRY_LOOKBACK, DELTA_TRY_CONV1 = range(5)

CFG_STRICT_HISTOGRAM = 1  # PCO_GFX_CFG_STRICT_HISTOGRAM: replay the reference's quickselect histogram pivot by pivot

DTYPE_BYTE = {"uint32": 1, "uint64": 2, "int32": 3, "int64": 4, "float32": 5, "float64": 6,
              "uint16": 7, "int16": 8, "float16": 9, "uint8": 10, "int8": 11}
DTYPE_BYTES = {1: 4, 2: 8, 3: 4, 4: 8, 5: 4, 6: 8, 7: 2, 8: 2, 9: 2, 10: 1, 11: 1}


class PcoChunkConfig(C.Structure):  # pco_c/src/lib.rs:21-32
    _fields_ = [("compression_level", C.c_uint), ("max_page_n", C.c_size_t)]


class PcoChunkConfigEx(C.Structure):
    _fields_ = [("compression_level", C.c_uint32), ("mode_kind", C.c_uint32), ("mode_f64", C.c_double),
                ("mode_u64", C.c_uint64), ("delta_kind", C.c_uint32), ("delta_order", C.c_uint32),
                ("max_page_n", C.c_uint64), ("enable_8_bit", C.c_uint32), ("flags", C.c_uint32)]


class EncodeTask(C.Structure):
    _fields_ = [("src", C.c_void_p), ("n", C.c_uint64), ("dst", C.c_void_p), ("dst_cap", C.c_uint64),
                ("dtype", C.c_uint32), ("reserved", C.c_uint32)]


class DecodeTask(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_uint64), ("dst", C.c_void_p), ("dst_cap", C.c_uint64),
                ("dtype", C.c_uint32), ("flags", C.c_uint32)]


class TaskResult(C.Structure):
    _fields_ = [("n_out", C.c_uint64), ("consumed", C.c_uint64), ("status", C.c_uint32), ("aux", C.c_uint32)]


KIND_NAMES = {ST_CORRUPTION: "Corruption", ST_INSUFFICIENT_DATA: "InsufficientData", ST_INVALID_ARGUMENT: "InvalidArgument",
              ST_UNSUPPORTED: "Unsupported", ST_DEVICE_ERROR: "DeviceError"}


class PageInfo(C.Structure):   # PcoGfxPageInfo: one piece (ChunkMeta or page) of a chunk written by pco_gfx_compress_wrapped_chunks
    _fields_ = [("offset", C.c_uint64), ("len", C.c_uint64), ("n", C.c_uint64), ("status", C.c_uint32), ("aux", C.c_uint32)]


class PageTask(C.Structure):   # PcoGfxPageTask: one wrapped page for pco_gfx_decompress_pages
    _fields_ = [("meta", C.c_void_p), ("meta_len", C.c_uint64), ("page", C.c_void_p), ("page_len", C.c_uint64),
                ("dst", C.c_void_p), ("page_n", C.c_uint64), ("dtype", C.c_uint32), ("format_major", C.c_uint32)]


class PcoGfxError(RuntimeError):
    """The reference's Python binding raises RuntimeError("pco error: pco <ErrorKind> error: <message>") (pco_python/src/utils.rs:78 over
    errors.rs:52-60): the same text here, so that callers (and the reference's own tests) that match on the kind keep working."""

    def __init__(self, code, status, msg):
        super().__init__(f"pco error: pco {KIND_NAMES.get(status, status)} error: {msg} (libpco_gfx code={code} status={status})")
        self.code, self.status = code, status


_lib = None


def lib():
    """Load libpco_gfx.so.  Raises if it has not been built (python -m pcodec_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m pcodec_amd.build` (needs hipcc). "
                              "pcodec_amd has no CPU fallback.")
        # When PyTorch-ROCm is used in the same process (device buffers for the batched API), load its HIP
        # runtime first so that both bind to ONE libamdhip64: two HIP runtimes in a process cannot both
        # open the device ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing
            pass
        L = C.CDLL(LIB_PATH)
        L.pco_gfx_last_error.restype = C.c_char_p
        L.pco_standalone_guarantee_file_size.restype = C.c_size_t
        L.pco_standalone_guarantee_file_size.argtypes = [C.c_size_t, C.c_ubyte]
        L.pco_gfx_guarantee_file_size.restype = C.c_size_t
        L.pco_gfx_guarantee_file_size.argtypes = [C.c_size_t, C.c_ubyte, C.c_uint64]
        L.pco_gfx_guarantee_chunk_size.restype = C.c_size_t
        L.pco_gfx_workspace_bytes.restype = C.c_size_t
        L.pco_gfx_workspace_bytes.argtypes = []
        L.pco_gfx_strict_histogram_fallbacks.restype = C.c_ulonglong
        L.pco_gfx_strict_histogram_fallbacks.argtypes = []
        L.pco_gfx_trail_givebacks.restype = C.c_ulonglong
        L.pco_gfx_trail_givebacks.argtypes = []
        L.pco_gfx_trail_marked.restype = C.c_ulonglong
        L.pco_gfx_trail_marked.argtypes = []
        L.pco_gfx_guarantee_chunk_size.argtypes = [C.c_size_t, C.c_ubyte]
        L.pco_standalone_simple_compress_into.argtypes = [C.c_void_p, C.c_size_t, C.c_ubyte, C.c_void_p, C.c_void_p,
                                                          C.c_size_t, C.POINTER(C.c_size_t)]
        L.pco_standalone_simple_decompress_into.argtypes = [C.c_void_p, C.c_size_t, C.c_ubyte, C.c_void_p, C.c_size_t,
                                                            C.POINTER(C.c_size_t)]
        L.pco_gfx_simple_compress_into_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_ubyte, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pco_gfx_compress_chunks.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pco_gfx_decompress_chunks.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pco_gfx_write_standalone_header.restype = C.c_size_t
        L.pco_gfx_write_standalone_header.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_ubyte]
        L.pco_gfx_write_standalone_footer.restype = C.c_size_t
        L.pco_gfx_write_standalone_footer.argtypes = [C.c_void_p, C.c_size_t]
        L.pco_wrapped_write_header.restype = C.c_size_t
        L.pco_wrapped_write_header.argtypes = [C.c_void_p, C.c_size_t]
        L.pco_gfx_wrapped_n_pages.restype = C.c_size_t
        L.pco_gfx_wrapped_n_pages.argtypes = [C.c_size_t, C.c_uint64]
        L.pco_gfx_wrapped_chunk_cap.restype = C.c_size_t
        L.pco_gfx_wrapped_chunk_cap.argtypes = [C.c_size_t, C.c_ubyte, C.c_void_p]
        L.pco_gfx_compress_wrapped_chunks.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pco_gfx_decompress_pages.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(code):
    if code != PcoSuccess:
        L = lib()
        raise PcoGfxError(code, L.pco_gfx_last_status(), (L.pco_gfx_last_error() or b"").decode())


def make_config(level=8, mode=MODE_AUTO, mode_f64=0.0, mode_u64=0, delta=DELTA_AUTO, delta_order=0, max_page_n=0,
                enable_8_bit=False, strict_histogram=False):
    return PcoChunkConfigEx(level, mode, mode_f64, mode_u64, delta, delta_order, max_page_n, 1 if enable_8_bit else 0,
                            CFG_STRICT_HISTOGRAM if strict_histogram else 0)
