"""ctypes loader for the parity ORACLE (oracle/libpco_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (pcodec_amd) never imports it.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libpco_oracle.so")

DTYPE_BYTE = {"u32": 1, "u64": 2, "i32": 3, "i64": 4, "f32": 5, "f64": 6,
              "u16": 7, "i16": 8, "f16": 9, "u8": 10, "i8": 11}
NP_DTYPE = {1: np.uint32, 2: np.uint64, 3: np.int32, 4: np.int64, 5: np.float32, 6: np.float64,
            7: np.uint16, 8: np.int16, 9: np.float16, 10: np.uint8, 11: np.int8}
NP_BITS_DTYPE = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}

# ModeSpecKind / DeltaSpecKind (shared numbering with include/pco_gfx.h)
MODE_AUTO, MODE_CLASSIC, MODE_TRY_FLOAT_MULT, MODE_TRY_FLOAT_QUANT, MODE_TRY_INT_MULT, MODE_TRY_DICT = range(6)
DELTA_AUTO, DELTA_NOOP, DELTA_TRY_CONSECUTIVE, DELTA_TRY_LOOKBACK, DELTA_TRY_CONV1 = range(5)

ERR_OK, ERR_CORRUPTION, ERR_INSUFFICIENT_DATA, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED = range(5)


class Config(C.Structure):
    _fields_ = [("compression_level", C.c_uint32), ("mode_kind", C.c_uint32), ("mode_f64", C.c_double),
                ("mode_u64", C.c_uint64), ("delta_kind", C.c_uint32), ("delta_order", C.c_uint32),
                ("max_page_n", C.c_uint64), ("enable_8_bit", C.c_uint32), ("reserved", C.c_uint32)]


def make_config(level=8, mode=MODE_AUTO, mode_f64=0.0, mode_u64=0, delta=DELTA_AUTO, delta_order=0,
                max_page_n=0, enable_8_bit=True):
    return Config(level, mode, mode_f64, mode_u64, delta, delta_order, max_page_n, 1 if enable_8_bit else 0, 0)


class ChunkInfo(C.Structure):
    _fields_ = [("mode_kind", C.c_uint32), ("mode_k", C.c_uint32), ("mode_base_latent", C.c_uint64),
                ("delta_kind", C.c_uint32), ("delta_order", C.c_uint32), ("window_n_log", C.c_uint32),
                ("state_n_log", C.c_uint32), ("n", C.c_uint32), ("dtype", C.c_uint32),
                ("var_present", C.c_uint32 * 3), ("ans_size_log", C.c_uint32 * 3), ("n_bins", C.c_uint32 * 3),
                ("standalone_version", C.c_uint32), ("uniform_type", C.c_uint32), ("fmt_major", C.c_uint32),
                ("fmt_minor", C.c_uint32), ("n_hint", C.c_uint64), ("meta_end_byte", C.c_uint64)]


class OracleError(Exception):
    def __init__(self, kind, msg):
        super().__init__(f"oracle error kind={kind}: {msg}")
        self.kind = kind


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.pco_oracle_last_error.restype = C.c_char_p
        _lib.pco_oracle_file_size_bound.restype = C.c_size_t
        _lib.pco_oracle_file_size_bound.argtypes = [C.c_size_t, C.c_uint8, C.c_uint64]
        _lib.pco_oracle_log2_approx.restype = C.c_float
        _lib.pco_oracle_log2_approx.argtypes = [C.c_float]
    return _lib


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().pco_oracle_last_error().decode())


def dtype_byte(arr):
    return DTYPE_BYTE[{"uint32": "u32", "uint64": "u64", "int32": "i32", "int64": "i64", "float32": "f32",
                       "float64": "f64", "uint16": "u16", "int16": "i16", "float16": "f16", "uint8": "u8",
                       "int8": "i8"}[arr.dtype.name]]


def file_size_bound(n, dt, max_page_n=0):
    return lib().pco_oracle_file_size_bound(n, dt, max_page_n)


def simple_compress(arr, config=None, uniform_type=False):
    arr = np.ascontiguousarray(arr)
    dt = dtype_byte(arr)
    cap = file_size_bound(arr.size, dt, config.max_page_n if config is not None else 0) + 64
    dst = np.empty(cap, dtype=np.uint8)
    n_written = C.c_size_t(0)
    rc = lib().pco_oracle_simple_compress(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dt),
                                          C.byref(config) if config is not None else None,
                                          C.c_int(1 if uniform_type else 0), dst.ctypes.data_as(C.c_void_p),
                                          C.c_size_t(cap), C.byref(n_written))
    _check(rc)
    return dst[: n_written.value].tobytes()


def simple_compress_exact(arr, config, exact_pages, uniform_type=False):
    """simple_compress under PagingSpec::Exact: one standalone chunk per entry of exact_pages (standalone/simple.rs:32-45)."""
    arr = np.ascontiguousarray(arr)
    dt = dtype_byte(arr)
    cap = file_size_bound(0, dt, 0) + sum(file_size_bound(max(int(p), 1), dt, 0) for p in exact_pages) + 64
    dst = np.empty(cap, dtype=np.uint8)
    n_written = C.c_size_t(0)
    ex = (C.c_size_t * max(len(exact_pages), 1))(*[int(x) for x in exact_pages])
    rc = lib().pco_oracle_simple_compress_exact(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dt), C.byref(config),
                                                C.c_int(1 if uniform_type else 0), ex, C.c_size_t(len(exact_pages)),
                                                dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n_written))
    _check(rc)
    return dst[: n_written.value].tobytes()


class TestEncSpec(C.Structure):
    """oracle/pco_oracle_testenc.hpp: the TEST-ONLY stream generator's spec."""
    __test__ = False
    _fields_ = [("mode_kind", C.c_uint32), ("delta_kind", C.c_uint32), ("mode_f64", C.c_double), ("mode_u64", C.c_uint64),
                ("order", C.c_uint32), ("secondary_uses_delta", C.c_uint32), ("window_n_log", C.c_uint32), ("state_n_log", C.c_uint32),
                ("lookback_seed", C.c_uint32), ("quantization", C.c_uint32), ("bias", C.c_int64), ("weights", C.c_int32 * 32),
                ("level", C.c_uint32), ("dict_first_appearance", C.c_uint32)]


TE_DELTA_NONE, TE_DELTA_CONSECUTIVE, TE_DELTA_LOOKBACK, TE_DELTA_CONV1 = range(4)


def test_encode(arr, chunks=None, mode=MODE_CLASSIC, mode_f64=0.0, mode_u64=0, delta=TE_DELTA_NONE, order=0, secondary_uses_delta=False,
                window_n_log=0, state_n_log=0, lookback_seed=0, quantization=0, bias=0, weights=(), level=8, dict_first_appearance=False):
    """A VALID standalone file written by the test-only generator (Dict mode, Conv1 delta, delta'd secondary, lookback state):
    one chunk per entry of `chunks`."""
    arr = np.ascontiguousarray(arr)
    chunks = [arr.size] if chunks is None else [int(c) for c in chunks]
    spec = TestEncSpec(mode, delta, mode_f64, mode_u64, order if delta != TE_DELTA_CONV1 else len(weights), 1 if secondary_uses_delta else 0,
                       window_n_log, state_n_log, lookback_seed, quantization, bias, (C.c_int32 * 32)(*[int(w) for w in weights]), level,
                       1 if dict_first_appearance else 0)
    dt = dtype_byte(arr)
    cap = 64 + arr.nbytes * 2 + 4096 * len(chunks) + 65536
    dst = np.empty(cap, np.uint8); n_written = C.c_size_t(0)
    cs = (C.c_size_t * len(chunks))(*chunks)
    rc = lib().pco_oracle_test_encode(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dt), C.byref(spec), cs, C.c_size_t(len(chunks)),
                                      dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n_written))
    _check(rc)
    return dst[: n_written.value].tobytes()


test_encode.__test__ = False


def simple_decompress(data, np_dtype, cap=None):
    dt = DTYPE_BYTE[{"uint32": "u32", "uint64": "u64", "int32": "i32", "int64": "i64", "float32": "f32",
                     "float64": "f64", "uint16": "u16", "int16": "i16", "float16": "f16", "uint8": "u8",
                     "int8": "i8"}[np.dtype(np_dtype).name]]
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if cap is None:
        cap = max(len(data) * 64, 1 << 20)
    out = np.empty(cap, dtype=np_dtype)
    n_written = C.c_size_t(0)
    rc = lib().pco_oracle_simple_decompress(buf.ctypes.data_as(C.c_void_p) if len(buf) else None, C.c_size_t(len(buf)),
                                            C.c_uint8(dt), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap),
                                            C.byref(n_written))
    _check(rc)
    return out[: n_written.value].copy()


def wrapped_compress(arr, config, max_pages=4096, exact_pages=None):
    """wrapped::ChunkCompressor on the oracle: (meta bytes, [page bytes], [page n]).  exact_pages: PagingSpec::Exact."""
    arr = np.ascontiguousarray(arr)
    n_pg = len(exact_pages) if exact_pages is not None else 0
    cap = file_size_bound(arr.size, dtype_byte(arr), int(config.max_page_n)) + 65536 + 64 * n_pg
    dst = np.empty(cap, np.uint8)
    sizes = (C.c_size_t * (max_pages + 1))(); page_ns = (C.c_size_t * max_pages)(); n_pages = C.c_size_t(0)
    if exact_pages is not None:
        ex = (C.c_size_t * max(n_pg, 1))(*[int(x) for x in exact_pages])
        rc = lib().pco_oracle_wrapped_compress_exact(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dtype_byte(arr)), C.byref(config), ex, C.c_size_t(n_pg),
                                                     dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), sizes, page_ns, C.c_size_t(max_pages), C.byref(n_pages))
    else:
        rc = lib().pco_oracle_wrapped_compress(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dtype_byte(arr)), C.byref(config),
                                               dst.ctypes.data_as(C.c_void_p), C.c_size_t(cap), sizes, page_ns, C.c_size_t(max_pages), C.byref(n_pages))
    _check(rc)
    pos = sizes[0]; meta = dst[:pos].tobytes(); pages = []
    for i in range(n_pages.value):
        pages.append(dst[pos: pos + sizes[1 + i]].tobytes()); pos += sizes[1 + i]
    return meta, pages, [int(page_ns[i]) for i in range(n_pages.value)]


def inspect_first_chunk(data, max_bins=4096):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    info = ChunkInfo()
    bins = np.zeros((3, max_bins, 3), dtype=np.uint64)
    rc = lib().pco_oracle_inspect_first_chunk(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(buf)), C.byref(info),
                                              bins.ctypes.data_as(C.c_void_p), C.c_size_t(max_bins))
    _check(rc)
    return info, [bins[v, : info.n_bins[v]] for v in range(3)]


def wrapped_page_prefix(meta, page, np_dtype, page_n, fmt_major=4):
    """wrapped::PageDecompressor batch after batch on a (possibly damaged) page: (numbers of the batches decoded before the first failure,
    error kind or 0, whether the page's own metadata already failed)."""
    dt = np.dtype(np_dtype)
    out = np.zeros(max(page_n, 1), dt)
    m = np.frombuffer(bytes(meta), np.uint8); p = np.frombuffer(bytes(page), np.uint8) if len(page) else np.zeros(1, np.uint8)
    n_ok = C.c_size_t(0); err = C.c_int(0); in_meta = C.c_int(0)
    rc = lib().pco_oracle_wrapped_page_prefix(m.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), p.ctypes.data_as(C.c_void_p), C.c_size_t(len(page)),
                                              C.c_uint8(dtype_byte(out)), C.c_uint8(fmt_major), C.c_size_t(page_n), out.ctypes.data_as(C.c_void_p),
                                              C.byref(n_ok), C.byref(err), C.byref(in_meta))
    _check(rc)
    return out[: n_ok.value], int(err.value), bool(in_meta.value)


def set_hist_rule(rule):
    """TEST HOOK of the oracle: 1 = every encode of this thread takes its histograms by the multiset rule (what the GPU computes), 0 = the
    reference's literal algorithm (default)."""
    lib().pco_oracle_set_hist_rule(C.c_int(rule))


def chunk_plan(arr, config, max_bins=4096):
    arr = np.ascontiguousarray(arr)
    info = ChunkInfo()
    bins = np.zeros((3, max_bins, 3), dtype=np.uint64)
    fb = C.c_int(0)
    rc = lib().pco_oracle_chunk_plan(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dtype_byte(arr)),
                                     C.byref(config) if config is not None else None, C.byref(info),
                                     bins.ctypes.data_as(C.c_void_p), C.c_size_t(max_bins), C.byref(fb))
    _check(rc)
    return info, [bins[v, : info.n_bins[v]] for v in range(3)], bool(fb.value)


def histogram(latents, n_bins_log, rule=0):
    latents = np.ascontiguousarray(latents)
    bits = latents.dtype.itemsize * 8
    cap = (1 << n_bins_log) + 1
    cnt = np.zeros(cap, np.uint64); lo = np.zeros(cap, np.uint64); hi = np.zeros(cap, np.uint64)
    n_out = C.c_size_t(0); fb = C.c_int(0)
    rc = lib().pco_oracle_histogram(latents.ctypes.data_as(C.c_void_p), C.c_size_t(latents.size), C.c_int(bits),
                                    C.c_uint32(n_bins_log), C.c_int(rule), cnt.ctypes.data_as(C.c_void_p),
                                    lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), C.byref(n_out),
                                    C.byref(fb))
    _check(rc)
    k = n_out.value
    return list(zip(cnt[:k].tolist(), lo[:k].tolist(), hi[:k].tolist())), bool(fb.value)


def optimize_bins(bins, latent_bits, ans_size_log):
    n = len(bins)
    cnt = np.array([b[0] for b in bins], np.uint64); lo = np.array([b[1] for b in bins], np.uint64)
    hi = np.array([b[2] for b in bins], np.uint64)
    ow = np.zeros(n, np.uint64); ol = np.zeros(n, np.uint64); ou = np.zeros(n, np.uint64); oo = np.zeros(n, np.uint32)
    n_out = C.c_size_t(0)
    rc = lib().pco_oracle_optimize_bins(cnt.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p),
                                        hi.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_int(latent_bits),
                                        C.c_uint32(ans_size_log), ow.ctypes.data_as(C.c_void_p),
                                        ol.ctypes.data_as(C.c_void_p), ou.ctypes.data_as(C.c_void_p),
                                        oo.ctypes.data_as(C.c_void_p), C.byref(n_out))
    _check(rc)
    k = n_out.value
    return list(zip(ow[:k].tolist(), ol[:k].tolist(), ou[:k].tolist(), oo[:k].tolist()))


def spread_state_symbols(weights):
    w = np.array(weights, np.uint32)
    total = int(w.sum()); size_log = total.bit_length() - 1
    out = np.zeros(total, np.uint32)
    _check(lib().pco_oracle_spread_state_symbols(C.c_uint32(size_log), w.ctypes.data_as(C.c_void_p), C.c_size_t(len(w)),
                                                 out.ctypes.data_as(C.c_void_p)))
    return out.tolist()


def quantize_weights(counts, total_count, max_size_log):
    c = np.array(counts, np.uint32); out = np.zeros(len(c), np.uint32); sl = C.c_uint32(0)
    _check(lib().pco_oracle_quantize_weights(c.ctypes.data_as(C.c_void_p), C.c_size_t(len(c)), C.c_size_t(total_count),
                                             C.c_uint32(max_size_log), out.ctypes.data_as(C.c_void_p), C.byref(sl)))
    return sl.value, out.tolist()


def quantize_weights_to(counts, total_count, size_log):
    c = np.array(counts, np.uint32); out = np.zeros(max(len(c), 1), np.uint32)
    _check(lib().pco_oracle_quantize_weights_to(c.ctypes.data_as(C.c_void_p), C.c_size_t(len(c)), C.c_size_t(total_count),
                                                C.c_uint32(size_log), out.ctypes.data_as(C.c_void_p)))
    return out.tolist()


def choose_lookbacks(latents, window_n_log, state_n_log=0):
    latents = np.ascontiguousarray(latents)
    out = np.zeros(max(latents.size, 1), np.uint32); n_out = C.c_size_t(0)
    _check(lib().pco_oracle_choose_lookbacks(latents.ctypes.data_as(C.c_void_p), C.c_size_t(latents.size),
                                             C.c_int(latents.dtype.itemsize * 8), C.c_uint32(window_n_log),
                                             C.c_uint32(state_n_log), out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[: n_out.value].copy()


def mode_sample_indices(n):
    out = np.zeros(n + 16, np.uint64); n_out = C.c_size_t(0)
    _check(lib().pco_oracle_mode_sample_indices(C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[: n_out.value].astype(np.int64)


def split_latents(arr, config):
    arr = np.ascontiguousarray(arr)
    bits = arr.dtype.itemsize * 8
    prim = np.zeros(arr.size, NP_BITS_DTYPE[bits]); sec = np.zeros(arr.size, NP_BITS_DTYPE[bits])
    mk = C.c_uint32(0); mp = C.c_uint64(0)
    _check(lib().pco_oracle_split_latents(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.c_uint8(dtype_byte(arr)),
                                          C.byref(config), prim.ctypes.data_as(C.c_void_p), sec.ctypes.data_as(C.c_void_p),
                                          C.byref(mk), C.byref(mp)))
    return prim, sec, mk.value, mp.value
