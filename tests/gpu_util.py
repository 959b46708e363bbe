"""Helpers for the -m gpu tests: synthetic BASELINE configs, device buffers, batched calls."""
import ctypes as C

import numpy as np

from pcodec_amd import _lib as G

N18 = 1 << 18


def cfg_pair(kind):
    """(product config, oracle config) for a BASELINE.json config name."""
    import oracle_lib as O
    table = {
        "c1": dict(mode=1, delta=1),                                    # Classic, NoOp
        "c2": dict(mode=1, delta=2, delta_order=1),                     # Classic, TryConsecutive(1)
        "c3": dict(mode=2, mode_f64=0.01, delta=1),                     # TryFloatMult(0.01), NoOp
        "c3d": dict(mode=2, mode_f64=0.01, delta=2, delta_order=1),
        "c4": dict(mode=1, delta=3),                                    # Classic, TryLookback
        "auto": dict(),
        "c2l12": dict(mode=1, delta=2, delta_order=1, level=12),   # BASELINE configs[1] at compression level 12 (up to 4096 bins)
    }[kind]
    return G.make_config(enable_8_bit=True, **table), O.make_config(**table)


def synth(kind, n=N18, seed=None):
    """SURVEY.md section 8(d) synthetic inputs."""
    if kind == "c1":
        return np.random.default_rng(1 if seed is None else seed).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if kind == "c2":
        r = np.random.default_rng(2 if seed is None else seed)
        return (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + r.integers(0, 512, n).astype(np.uint64))
    if kind in ("c3", "c3d"):
        return np.random.default_rng(3 if seed is None else seed).integers(1000, 10000, n) / 100.0
    if kind == "c4":
        base = np.random.default_rng(40).integers(-(1 << 40), 1 << 40, 365)
        return (base[np.arange(n) % 365] + np.random.default_rng(4 if seed is None else seed).integers(-3, 4, n)).astype(np.int64)
    raise KeyError(kind)


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    u = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(u), b.view(u))


def gpu_simple_decompress(data, np_dtype, cap):
    L = G.lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(max(cap, 1), dtype=np_dtype)
    n = C.c_size_t(0)
    code = L.pco_standalone_simple_decompress_into(buf.ctypes.data_as(C.c_void_p) if len(buf) else None, len(buf),
                                                   G.DTYPE_BYTE[np.dtype(np_dtype).name], out.ctypes.data_as(C.c_void_p),
                                                   cap, C.byref(n))
    G.check(code)
    return out[: n.value].copy()


def gpu_simple_compress(arr, cfg, uniform_type=False):
    L = G.lib()
    arr = np.ascontiguousarray(arr)
    dt = G.DTYPE_BYTE[arr.dtype.name]
    cap = L.pco_gfx_guarantee_file_size(arr.size, dt, cfg.max_page_n) + 64
    dst = np.empty(cap, np.uint8); n = C.c_size_t(0)
    code = L.pco_gfx_simple_compress_into_ex(arr.ctypes.data_as(C.c_void_p), arr.size, dt, C.byref(cfg),
                                             1 if uniform_type else 0, dst.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    G.check(code)
    return dst[: n.value].tobytes()


def gpu_batched(arrays, cfg):
    """One pco_gfx_compress_chunks call over `arrays` (any mix of dtypes / sizes; device buffers through torch), then one
    pco_gfx_decompress_chunks call over what it produced.  Returns ([standalone chunk bytes], [decoded arrays])."""
    import torch
    L = G.lib()
    arrays = [np.ascontiguousarray(a) for a in arrays]
    srcs = [torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() for a in arrays]
    caps = [(L.pco_gfx_guarantee_chunk_size(a.size, G.DTYPE_BYTE[a.dtype.name]) + 64 + 15) // 16 * 16 for a in arrays]
    dsts = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
    k = len(arrays)
    tasks = (G.EncodeTask * k)(*[G.EncodeTask(s.data_ptr(), a.size, d.data_ptr(), c, G.DTYPE_BYTE[a.dtype.name], 0)
                                 for a, s, d, c in zip(arrays, srcs, dsts, caps)])
    res = (G.TaskResult * k)()
    G.check(L.pco_gfx_compress_chunks(k, tasks, C.byref(cfg), res, None, None))
    outs = [torch.empty(max(a.nbytes, 1), dtype=torch.uint8, device="cuda") for a in arrays]
    dtasks = (G.DecodeTask * k)(*[G.DecodeTask(d.data_ptr(), res[i].n_out, o.data_ptr(), a.size, G.DTYPE_BYTE[a.dtype.name], 0)
                                  for i, (a, d, o) in enumerate(zip(arrays, dsts, outs))])
    dres = (G.TaskResult * k)()
    G.check(L.pco_gfx_decompress_chunks(k, dtasks, dres, None, None))
    chunks = [bytes(dsts[i][: res[i].n_out].cpu().numpy()) for i in range(k)]
    back = []
    for i, a in enumerate(arrays):
        assert dres[i].n_out == a.size and dres[i].consumed == res[i].n_out, (i, dres[i].n_out, a.size)
        back.append(outs[i][: a.nbytes].cpu().numpy().view(a.dtype))
    return chunks, back


def chunk_of_file(file_bytes, chunk_len):
    """The single chunk inside an oracle-written one-chunk standalone file (header | chunk | 0x00)."""
    return file_bytes[len(file_bytes) - 1 - chunk_len:-1]


def profile_names(L):
    """Names of the kernels / spans timed since pco_gfx_profile_begin() (pco_gfx_profile_end), as a sorted list without repeats."""
    import ctypes as C
    import torch
    torch.cuda.synchronize()
    names = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names, len(names), ms, 4096)
    raw = names.raw; out = []; pos = 0
    for _ in range(nk):
        e = raw.index(b"\0", pos); out.append(raw[pos:e].decode()); pos = e + 1
    return sorted(set(out))

