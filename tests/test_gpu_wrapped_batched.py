"""-m gpu: the wrapped surface, batched, on device buffers (include/pco_gfx.h section 4b: pco_gfx_compress_wrapped_chunks /
pco_gfx_decompress_pages) -- what the reference's `pcopage` bench codec does per chunk (pco_cli/src/bench/codecs/pcopage.rs:33-113:
ChunkCompressor::write_meta, write_page per page; PageDecompressor::read per page), for many chunks in one call.  Every ChunkMeta and
every page is compared byte for byte with the oracle's wrapped::ChunkCompressor (wrapped/chunk_compressor.rs:564,659-705), and every page
is decoded from its own buffer beside its chunk's ChunkMeta (wrapped/chunk_decompressor.rs:74-80)."""
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
    return lib


def wrapped_batch(L, arrays, cfg):
    """pco_gfx_compress_wrapped_chunks over `arrays`: ([(meta bytes, [page bytes], [page_n])], device state for the decode)."""
    import torch
    arrays = [np.ascontiguousarray(a) for a in arrays]
    k = len(arrays)
    srcs = [torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() for a in arrays]
    caps = [L.pco_gfx_wrapped_chunk_cap(a.size, G.DTYPE_BYTE[a.dtype.name], C.addressof(cfg)) for a in arrays]
    dsts = [torch.zeros(c + 64, dtype=torch.uint8, device="cuda") for c in caps]
    assert all(d.data_ptr() % 16 == 0 for d in dsts)
    tasks = (G.EncodeTask * k)(*[G.EncodeTask(s.data_ptr(), a.size, d.data_ptr(), c, G.DTYPE_BYTE[a.dtype.name], 0) for a, s, d, c in zip(arrays, srcs, dsts, caps)])
    n_pages = [L.pco_gfx_wrapped_n_pages(a.size, cfg.max_page_n) for a in arrays]
    infos = (G.PageInfo * (sum(n_pages) + k))()
    G.check(L.pco_gfx_compress_wrapped_chunks(k, tasks, C.addressof(cfg), infos, None))
    out = []; at = 0
    for i, a in enumerate(arrays):
        host = dsts[i].cpu().numpy()
        m = infos[at]; assert m.status == 0 and m.n == 0 and m.offset == 0
        meta = bytes(host[: m.len]); pages = []; ns = []
        for p in range(n_pages[i]):
            e = infos[at + 1 + p]; assert e.status == 0
            assert e.offset % 16 == 0
            pages.append(bytes(host[e.offset: e.offset + e.len])); ns.append(int(e.n))
        out.append((meta, pages, ns)); at += 1 + n_pages[i]
    return out, (dsts, infos, n_pages)


def decode_pages(L, arrays, state, order=None, fmt=4):
    """Every page of every chunk through ONE pco_gfx_decompress_pages call, each page into its slice of the chunk's output."""
    import torch
    dsts, infos, n_pages = state
    outs = [torch.zeros(max(a.nbytes, 1) + 64, dtype=torch.uint8, device="cuda") for a in arrays]
    tasks = []; at = 0
    for i, a in enumerate(arrays):
        m = infos[at]; start = 0
        for p in range(n_pages[i]):
            e = infos[at + 1 + p]
            tasks.append(G.PageTask(dsts[i].data_ptr(), m.len, dsts[i].data_ptr() + e.offset, e.len, outs[i].data_ptr() + start * a.dtype.itemsize, e.n, G.DTYPE_BYTE[a.dtype.name], fmt))
            start += e.n
        at += 1 + n_pages[i]
    if order is not None:
        tasks = [tasks[j] for j in order]
    arr = (G.PageTask * len(tasks))(*tasks)
    res = (G.TaskResult * len(tasks))()
    code = L.pco_gfx_decompress_pages(len(tasks), arr, res, None, None)
    return code, res, [o[: a.nbytes].cpu().numpy().view(a.dtype) for o, a in zip(outs, arrays)], tasks


CASES = {
    "c2": (dict(mode=1, delta=2, delta_order=1), lambda n, s: U.synth("c2", n, seed=s)),
    "c2_order3": (dict(mode=1, delta=2, delta_order=3), lambda n, s: U.synth("c2", n, seed=s)),
    "c3": (dict(mode=2, mode_f64=0.01, delta=1), lambda n, s: U.synth("c3", n, seed=s)),
    "c4": (dict(mode=1, delta=3), lambda n, s: U.synth("c4", n, seed=s)),
    "c1": (dict(mode=1, delta=1), lambda n, s: U.synth("c1", n, seed=s)),
    "auto_i32": (dict(), lambda n, s: (np.random.default_rng(s).normal(size=n) * 1000).astype(np.int32) * 10),
    "auto_f32": (dict(), lambda n, s: np.random.default_rng(s).normal(size=n).astype(np.float32)),
    "quant_f64": (dict(mode=3, mode_u64=20, delta=1), lambda n, s: np.random.default_rng(s).normal(size=n)),
    "i16": (dict(mode=1, delta=2, delta_order=1), lambda n, s: np.cumsum(np.random.default_rng(s).integers(-5, 6, n)).astype(np.int16)),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("max_page_n", (700, 4096, 1 << 18))
def test_wrapped_batch_matches_the_oracle_and_round_trips(L, case, max_page_n):
    kw, gen = CASES[case]
    sizes = [1, 255, 256, 257, 5000, 4096 * 3, 20001]
    arrays = [gen(n, 100 + i) for i, n in enumerate(sizes)]
    cfg = G.make_config(max_page_n=max_page_n, enable_8_bit=True, **kw)
    got, state = wrapped_batch(L, arrays, cfg)
    for a, (meta, pages, ns) in zip(arrays, got):
        want = O.wrapped_compress(a, O.make_config(max_page_n=max_page_n, **kw))
        assert ns == want[2]
        assert meta == want[0], f"{case}: ChunkMeta differs from the oracle's (n = {a.size})"
        assert pages == want[1], f"{case}: page bytes differ from the oracle's (n = {a.size})"
    n_tasks = sum(state[2])
    order = np.random.default_rng(7).permutation(n_tasks)   # pages are independent: any order, any mix of chunks
    code, res, back, tasks = decode_pages(L, arrays, state, order=order)
    assert code == G.PcoSuccess
    for j in range(n_tasks):
        assert res[j].status == 0 and res[j].n_out == tasks[j].page_n and res[j].consumed == tasks[j].page_len
    for a, b in zip(arrays, back):
        assert U.bits_equal(a, b)


def test_many_pages_take_the_walkers_and_the_expanders_under_them(L):
    """64 chunks of 2^16 numbers in pages of 4096: 1024 page tasks in one call -- the publishing walker + the expanders under it."""
    arrays = [U.synth("c2", 1 << 16, seed=s) for s in range(64)]
    cfg = G.make_config(mode=1, delta=2, delta_order=1, max_page_n=4096)
    got, state = wrapped_batch(L, arrays, cfg)
    for a, (meta, pages, ns) in list(zip(arrays, got))[:6]:
        assert (meta, pages, ns) == O.wrapped_compress(a, O.make_config(mode=1, delta=2, delta_order=1, max_page_n=4096))
    L.pco_gfx_profile_begin()
    marked0 = L.pco_gfx_trail_marked()
    code, res, back, tasks = decode_pages(L, arrays, state)
    names = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names, len(names), ms, 4096)
    kernels = set(names.raw.split(b"\0")[:nk])
    assert code == G.PcoSuccess and all(U.bits_equal(a, b) for a, b in zip(arrays, back))
    assert any(k.startswith(b"dec_walk+trail") for k in kernels), kernels
    assert L.pco_gfx_trail_marked() - marked0 == 1024 and L.pco_gfx_trail_givebacks() == 0   # every page went to an expander wave and stayed there


def test_a_damaged_page_fails_alone_with_the_reference_timing(L):
    """One truncated page and one with a flipped bit among healthy ones: those tasks report what the oracle's page decoder reports (and how many
    numbers came out before the bad batch), every other page of the call decodes."""
    import torch
    arrays = [U.synth("c2", 6000, seed=s) for s in range(3)]
    cfg = G.make_config(mode=1, delta=2, delta_order=1, max_page_n=2000)
    got, state = wrapped_batch(L, arrays, cfg)
    dsts, infos, n_pages = state
    code, res, back, tasks = decode_pages(L, arrays, state)
    assert code == G.PcoSuccess
    # truncate page 1 of chunk 0 to 60 % of its bytes
    t = list(tasks)
    cut = int(t[1].page_len * 0.6)
    short = torch.zeros(cut + 64, dtype=torch.uint8, device="cuda"); short[:cut] = dsts[0][infos[2].offset: infos[2].offset + cut]
    t[1] = G.PageTask(t[1].meta, t[1].meta_len, short.data_ptr(), cut, t[1].dst, t[1].page_n, t[1].dtype, 4)
    arr = (G.PageTask * len(t))(*t); res = (G.TaskResult * len(t))()
    code = L.pco_gfx_decompress_pages(len(t), arr, res, None, None)
    assert code == G.PcoDecompressionError and L.pco_gfx_last_status() == G.ST_INSUFFICIENT_DATA
    meta, pages, ns = got[0]
    ok_nums, status, in_meta = O.wrapped_page_prefix(meta, pages[1][:cut], arrays[0].dtype, ns[1])
    assert res[1].status == G.ST_INSUFFICIENT_DATA == status and not in_meta and res[1].n_out == len(ok_nums) > 0 and (res[1].aux & 1)
    assert all(res[j].status == 0 for j in range(len(t)) if j != 1)


def test_argument_errors(L):
    import torch
    x = U.synth("c2", 3000)
    cfg = G.make_config(mode=1, delta=2, delta_order=1, max_page_n=1000)
    src = torch.from_numpy(x.view(np.uint8).copy()).cuda()
    cap = L.pco_gfx_wrapped_chunk_cap(x.size, 2, C.addressof(cfg))
    assert L.pco_gfx_wrapped_n_pages(3000, 1000) == 3 and L.pco_gfx_wrapped_n_pages(3001, 1000) == 4 and L.pco_gfx_wrapped_n_pages(5, 0) == 1
    dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
    infos = (G.PageInfo * 4)()
    task = (G.EncodeTask * 1)(G.EncodeTask(src.data_ptr(), x.size, dst.data_ptr(), cap - 16, 2, 0))
    assert L.pco_gfx_compress_wrapped_chunks(1, task, C.addressof(cfg), infos, None) == G.PcoCompressionError and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    task = (G.EncodeTask * 1)(G.EncodeTask(src.data_ptr(), x.size, dst.data_ptr() + 8, cap, 2, 0))
    assert L.pco_gfx_compress_wrapped_chunks(1, task, C.addressof(cfg), infos, None) == G.PcoCompressionError and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    pt = (G.PageTask * 1)(G.PageTask(dst.data_ptr(), 10, dst.data_ptr(), 10, dst.data_ptr(), 5, 2, 5))
    res = (G.TaskResult * 1)()
    assert L.pco_gfx_decompress_pages(1, pt, res, None, None) == G.PcoDecompressionError and L.pco_gfx_last_status() == G.ST_CORRUPTION
