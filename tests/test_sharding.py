"""N > 1 path on CPU: world-size-2 gloo run of the chunk sharding + page gather.
The per-chunk codec here is the oracle (test infrastructure); on the GPU box bench.py --gather runs the same
sharding code over RCCL with libpco_gfx as the codec."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_lib as O
from pcodec_amd import sharding as S


def test_shard_range_partitions_in_order():
    for n in [0, 1, 5, 8, 1023, 1024]:
        for world in [1, 2, 3, 8]:
            seen = []
            for r in range(world):
                a, b = S.shard_range(n, r, world)
                assert 0 <= a <= b <= n and b - a in (n // world, n // world + 1)
                seen += list(range(a, b))
                for c in range(a, b):
                    assert S.shard_of_chunk(c, n, world) == r
            assert seen == list(range(n))


def _chunks(n_chunks, chunk_n):
    rng = np.random.default_rng(11)
    return [(np.uint64(1 << 40) + np.uint64(1000) * np.arange(chunk_n, dtype=np.uint64) + rng.integers(0, 512, chunk_n).astype(np.uint64)) for _ in range(n_chunks)]


def _chunk_bytes(a, cfg):
    whole = O.simple_compress(a, cfg)           # header | chunk | 0x00
    _, _ = O.inspect_first_chunk(whole)
    hdr = _header_len(whole)
    return whole[hdr:-1]


def _header_len(whole):
    # "pco!" ver uniform varint(n_hint) pad fmt_major fmt_minor (standalone/compressor.rs:85-113)
    assert whole[:4] == b"pco!" and whole[4] == 3
    bits = int.from_bytes(whole[6:16], "little")
    power = (bits & 63) + 1
    return 6 + (6 + power + 7) // 8 + 2


def _worker(rank, world, port, n_chunks, chunk_n, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = O.make_config(mode=1, delta=2, delta_order=1)
        data = _chunks(n_chunks, chunk_n)
        a, b = S.shard_range(n_chunks, rank, world)
        mine = [_chunk_bytes(data[c], cfg) for c in range(a, b)]
        payload, sizes = S.pack_chunks(mine)
        bufs, all_sizes = S.gather_pages(torch.from_numpy(payload), torch.from_numpy(sizes), dst=0)
        assert [int(s.numel()) for s in all_sizes] == [S.shard_range(n_chunks, r, world)[1] - S.shard_range(n_chunks, r, world)[0] for r in range(world)]
        if rank == 0:
            q.put([bytes(x.numpy()) for x in bufs])
        # decode direction: the root scatters the byte ranges of the assembled stream, every rank decodes its own block
        totals = [int(s.sum().item()) for s in all_sizes]
        offs = [0]
        for t in totals: offs.append(offs[-1] + t)
        stream = torch.cat(bufs) if rank == 0 else None
        recv = torch.zeros(totals[rank] + 16, dtype=torch.uint8)
        got = S.scatter_stream(stream, offs, recv, src=0)
        assert got == totals[rank] and bytes(recv[:got].numpy()) == b"".join(mine)
        hdr = O.simple_compress(data[0], cfg)[: _header_len(O.simple_compress(data[0], cfg))]
        pos = 0
        for k, c in enumerate(range(a, b)):
            sz = int(all_sizes[rank][k]); blob = hdr + bytes(recv[pos: pos + sz].numpy()) + b"\x00"; pos += sz
            assert np.array_equal(O.simple_decompress(blob, np.uint64, cap=chunk_n + 8), data[c])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks,world", [(5, 2), (7, 3)])
def test_gather_reassembles_the_file_and_scatter_feeds_the_decoders(n_chunks, world):
    import torch.multiprocessing as mp
    chunk_n = 3000
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chunks, chunk_n, q)) for r in range(world)]
    for p in procs: p.start()
    per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    # reference result: all chunks compressed by one process, in order
    cfg = O.make_config(mode=1, delta=2, delta_order=1)
    data = _chunks(n_chunks, chunk_n)
    single = [_chunk_bytes(a, cfg) for a in data]
    assert b"".join(per_rank) == b"".join(single)
    # the assembled standalone file decodes to the concatenated input
    from pcodec_amd import _lib as G
    L = G.lib()
    hdr = np.zeros(32, np.uint8)
    k = L.pco_gfx_write_standalone_header(hdr.ctypes.data_as(C.c_void_p), 32, n_chunks * chunk_n, 0)
    blob = S.assemble_standalone_file(bytes(hdr[:k]), per_rank)
    back = O.simple_decompress(blob, np.uint64, cap=n_chunks * chunk_n + 8)
    assert np.array_equal(back, np.concatenate(data))


def _nccl_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import gpu_util as U
        from pcodec_amd import _lib as G
        rng = np.random.default_rng(100 + rank)
        arrays = [(np.uint64(1 << 40) + np.uint64(1000) * np.arange(5000, dtype=np.uint64) + rng.integers(0, 512, 5000).astype(np.uint64)) for _ in range(6 + rank)]
        kw = dict(mode=1, delta=2, delta_order=1)
        chunks, _ = U.gpu_batched(arrays, G.make_config(**kw))
        payload = torch.from_numpy(np.frombuffer(b"".join(chunks), np.uint8).copy()).cuda()
        out, offs = S.gather_stream(payload, payload.numel(), dst=0)
        recv = torch.zeros(payload.numel() + 16, dtype=torch.uint8, device="cuda")
        got = S.scatter_stream(out, offs, recv, src=0)
        assert got == payload.numel() and torch.equal(recv[:got], payload)
        if rank == 0:
            q.put(bytes(out[: offs[-1]].cpu().numpy()))
        else:
            q.put(b"".join(chunks))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_rccl_gather_and_scatter_of_chunk_bytes():
    """The same gather-v / scatter over RCCL with libpco_gfx as the codec; needs two GPUs (the round's GPU box has one: skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    parts = [q.get(timeout=300), q.get(timeout=300)]
    for p in procs:
        p.join(timeout=300); assert p.exitcode == 0
    whole = max(parts, key=len); tail = min(parts, key=len)
    assert whole.endswith(tail)      # rank 0's gathered stream = its own chunks followed by rank 1's


# ------------------------------------------------------------------------------------------------ C ABI over RCCL (include/pco_gfx.h section 5)
def _comm_api(L):
    L.pco_gfx_comm_unique_id.argtypes = [C.c_void_p]
    L.pco_gfx_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pco_gfx_comm_free.argtypes = [C.c_void_p]
    L.pco_gfx_gather_chunks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.pco_gfx_scatter_chunks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.pco_gfx_compact_chunks.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]


def _rank_round_trip(L, comm, rank, world, chunks_of_rank, cfg_kw, root=0):
    """One rank's part of the C-ABI file assembly: encode my block of chunks, compact, pco_gfx_gather_chunks to `root`,
    pco_gfx_scatter_chunks back, decode.  Returns (file body on the root else None, my decoded arrays)."""
    import torch
    from pcodec_amd import _lib as G
    arrays = chunks_of_rank
    k = len(arrays)
    srcs = [torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() for a in arrays]
    caps = [(L.pco_gfx_guarantee_chunk_size(a.size, G.DTYPE_BYTE[a.dtype.name]) + 64 + 15) // 16 * 16 for a in arrays]
    dsts = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
    tasks = (G.EncodeTask * max(k, 1))(*[G.EncodeTask(s.data_ptr(), a.size, d.data_ptr(), c, G.DTYPE_BYTE[a.dtype.name], 0) for a, s, d, c in zip(arrays, srcs, dsts, caps)])
    res = (G.TaskResult * max(k, 1))()
    d_res = torch.zeros(max(k, 1) * C.sizeof(G.TaskResult), dtype=torch.uint8, device="cuda")
    cfg = G.make_config(**cfg_kw)
    if k:
        G.check(L.pco_gfx_compress_chunks(k, tasks, C.byref(cfg), res, d_res.data_ptr(), None))
    stream_cap = sum(caps) + 64
    payload = torch.zeros(stream_cap, dtype=torch.uint8, device="cuda")
    d_offs = torch.zeros(k + 1, dtype=torch.int64, device="cuda")
    total = C.c_uint64(0)
    G.check(L.pco_gfx_compact_chunks(k, tasks, d_res.data_ptr(), payload.data_ptr(), stream_cap - 64, 0, d_offs.data_ptr(), C.byref(total), None))
    offsets = (C.c_uint64 * (world + 1))()
    file_cap = 1 << 26
    file_body = torch.zeros(file_cap if rank == root else 16, dtype=torch.uint8, device="cuda")
    G.check(L.pco_gfx_gather_chunks(comm, root, payload.data_ptr(), total.value, file_body.data_ptr(), file_cap, 0, offsets, None))
    torch.cuda.synchronize()
    assert offsets[rank + 1] - offsets[rank] == total.value
    recv = torch.zeros(stream_cap, dtype=torch.uint8, device="cuda")
    got = C.c_uint64(0)
    G.check(L.pco_gfx_scatter_chunks(comm, root, file_body.data_ptr(), 0, offsets, recv.data_ptr(), stream_cap - 16, C.byref(got), None))
    torch.cuda.synchronize()
    assert got.value == total.value and torch.equal(recv[: got.value], payload[: got.value])
    sizes = np.array([res[i].n_out for i in range(k)], dtype=np.uint64)
    starts = np.concatenate([[0], np.cumsum(sizes[:-1])]).astype(np.uint64) if k else np.zeros(0, np.uint64)
    outs = [torch.empty(max(a.nbytes, 1), dtype=torch.uint8, device="cuda") for a in arrays]
    dtasks = (G.DecodeTask * max(k, 1))(*[G.DecodeTask(recv.data_ptr() + int(starts[i]), int(sizes[i]), outs[i].data_ptr(), arrays[i].size, G.DTYPE_BYTE[arrays[i].dtype.name], 0) for i in range(k)])
    dres = (G.TaskResult * max(k, 1))()
    if k:
        G.check(L.pco_gfx_decompress_chunks(k, dtasks, dres, None, None))
    back = [outs[i][: arrays[i].nbytes].cpu().numpy().view(arrays[i].dtype) for i in range(k)]
    body = bytes(file_body[: offsets[world]].cpu().numpy()) if rank == root else None
    return body, back


def _sharded_inputs(n_chunks):
    rng = np.random.default_rng(7)
    return [(np.uint64(1 << 40) + np.uint64(1000) * np.arange(3000 + 97 * i, dtype=np.uint64) + rng.integers(0, 512, 3000 + 97 * i).astype(np.uint64)) for i in range(n_chunks)]


@pytest.mark.gpu
def test_c_abi_gather_scatter_single_rank_executes_rccl():
    """include/pco_gfx.h section 5 with a communicator of ONE rank: ncclGetUniqueId, ncclCommInitRank, the size all-gather and the
    local copy all run (a box with one GPU cannot hold two ranks: RCCL refuses duplicate devices); the assembled body is the oracle's
    file minus header and terminator."""
    import torch
    import oracle_lib as O
    from pcodec_amd import _lib as G
    L = G.lib(); _comm_api(L)
    assert L.pco_gfx_device_count() >= 1
    ident = (C.c_ubyte * 128)()
    G.check(L.pco_gfx_comm_unique_id(ident))
    comm = C.c_void_p()
    G.check(L.pco_gfx_comm_init(ident, 1, 0, C.byref(comm)))
    try:
        arrays = _sharded_inputs(5)
        kw = dict(mode=1, delta=2, delta_order=1)
        body, back = _rank_round_trip(L, comm, 0, 1, arrays, kw)
        want = b"".join(U_chunk(O, a, kw) for a in arrays)
        assert body == want
        for a, b in zip(arrays, back):
            assert np.array_equal(a, b)
    finally:
        L.pco_gfx_comm_free(comm)
    del torch


def U_chunk(O, a, kw):
    """The standalone chunk the oracle writes for `a`: its one-chunk file minus header and terminator."""
    f = O.simple_compress(a, O.make_config(**kw))
    return f[_header_len(f): -1]


def _header_len(f):
    """standalone/compressor.rs:85-105: magic(4) version(1) uniform type(1) varint n_hint, byte aligned, then the 2 format bytes."""
    bits = int.from_bytes(f[6:16], "little")
    power = 1 + (bits & 63)
    return 6 + (6 + power + 7) // 8 + 2


def _block_of(n_chunks, rank, world, empty_rank):
    """Contiguous chunk blocks as pcodec_amd.sharding deals them, with one rank (optionally) owning nothing: the ranks left share the chunks."""
    from pcodec_amd.sharding import shard_range
    if empty_rank is None:
        return shard_range(n_chunks, rank, world)
    if rank == empty_rank:
        return 0, 0
    return shard_range(n_chunks, rank - (1 if rank > empty_rank else 0), world - 1)


def _c_abi_rank_main(rank, world, id_path, out_path, root=0, one_device=False, empty_rank=None, failure=None):
    import torch
    from pcodec_amd import _lib as G
    torch.cuda.set_device(0 if one_device else rank)
    L = G.lib(); _comm_api(L)
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        G.check(L.pco_gfx_comm_unique_id(ident))
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(id_path + ".tmp", id_path)
    else:
        import time
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.1)
        C.memmove(ident, open(id_path, "rb").read(), 128)
    comm = C.c_void_p()
    G.check(L.pco_gfx_comm_init(ident, world, rank, C.byref(comm)))
    arrays = _sharded_inputs(11)
    c0, c1 = _block_of(len(arrays), rank, world, empty_rank)
    kw = dict(mode=1, delta=2, delta_order=1)
    if failure is not None:
        ok = _collective_failures(L, comm, rank, world, root)
        body = None
    else:
        body, back = _rank_round_trip(L, comm, rank, world, arrays[c0:c1], kw, root=root)
        ok = all(np.array_equal(a, b) for a, b in zip(arrays[c0:c1], back))
    L.pco_gfx_comm_free(comm)
    with open(f"{out_path}.{rank}", "wb") as f:
        f.write(b"OK" if ok else b"NO")
        if body is not None:
            f.write(body)


def _collective_failures(L, comm, rank, world, root):
    """A rank that cannot take part must make EVERY rank return INVALID_ARGUMENT -- before anything is posted, so nobody hangs
    (the calls below would block for ever otherwise) -- and the communicator must still work afterwards."""
    import torch
    from pcodec_amd import _lib as G
    mine = 1000 + 100 * rank
    payload = torch.full((mine + 64,), rank + 1, dtype=torch.uint8, device="cuda")
    offsets = (C.c_uint64 * (world + 1))()
    small = torch.zeros(64, dtype=torch.uint8, device="cuda")
    # 1. the root's file buffer is too small
    rc = L.pco_gfx_gather_chunks(comm, root, payload.data_ptr(), mine, small.data_ptr(), 64, 0, offsets, None)
    ok = rc != 0 and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT and offsets[world] == sum(1000 + 100 * r for r in range(world))
    # 2. the root passes no buffer at all
    rc = L.pco_gfx_gather_chunks(comm, root, payload.data_ptr(), mine, None if rank == root else small.data_ptr(), 1 << 20, 0, offsets, None)
    ok = ok and rc != 0 and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    # 3. a good gather, then a scatter in which ONE non-root rank has too little room
    big = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    G.check(L.pco_gfx_gather_chunks(comm, root, payload.data_ptr(), mine, big.data_ptr(), 1 << 16, 0, offsets, None))
    torch.cuda.synchronize()
    if rank == root:
        want = b"".join(bytes([r + 1]) * (1000 + 100 * r) for r in range(world))
        ok = ok and bytes(big[: offsets[world]].cpu().numpy()) == want
    victim = (root + 1) % world
    recv = torch.zeros(1 << 14, dtype=torch.uint8, device="cuda")
    got = C.c_uint64(0)
    rc = L.pco_gfx_scatter_chunks(comm, root, big.data_ptr(), 0, offsets, recv.data_ptr(), 10 if rank == victim else 1 << 14, C.byref(got), None)
    ok = ok and rc != 0 and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    # 4. per-rank argument errors travel with the size exchange too (round 4 threw them locally and left the peers in the all-gather):
    #    one rank without an offsets table, one rank with bytes but no buffer, one rank naming a root that does not exist -- in the gather
    #    and in the scatter
    bad = (root + 1) % world
    offs2 = (C.c_uint64 * (world + 1))()
    for variant in range(3):
        a_off = None if (variant == 0 and rank == bad) else offs2
        a_buf = None if (variant == 1 and rank == bad) else payload.data_ptr()
        a_root = world + 3 if (variant == 2 and rank == bad) else root
        rc = L.pco_gfx_gather_chunks(comm, a_root, a_buf, mine, big.data_ptr(), 1 << 16, 0, a_off, None)
        ok = ok and rc != 0 and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    rc = L.pco_gfx_scatter_chunks(comm, root, big.data_ptr(), 0, None if rank == bad else offsets, recv.data_ptr(), 1 << 14, C.byref(got), None)
    ok = ok and rc != 0 and L.pco_gfx_last_status() == G.ST_INVALID_ARGUMENT
    # 5. and the communicator still works
    G.check(L.pco_gfx_scatter_chunks(comm, root, big.data_ptr(), 0, offsets, recv.data_ptr(), 1 << 14, C.byref(got), None))
    torch.cuda.synchronize()
    ok = ok and got.value == mine and bool((recv[:mine] == rank + 1).all())
    return bool(ok)


def _fake_rccl_lib():
    """tests/fake_rccl.so: the loopback transport (tests/fake_rccl.cpp); built by __graft_entry__.build(), or here when it is missing."""
    import fake_rccl_build
    return fake_rccl_build.build()


def _run_ranks(tmp_path, world, extra, env_extra=None):
    import subprocess
    idp, outp = str(tmp_path / "id"), str(tmp_path / "out")
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, "-c", f"import sys; sys.path.insert(0, {HERE!r}); sys.path.insert(0, {os.path.join(HERE, '..')!r}); import test_sharding as T; T._c_abi_rank_main({r}, {world}, {idp!r}, {outp!r}, {extra})"], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [open(f"{outp}.{r}", "rb").read() for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,root,empty_rank", [(2, 0, None), (2, 1, None), (3, 2, 1), (3, 0, 0), (4, 1, 3)])
def test_c_abi_gather_scatter_of_several_ranks_on_one_device(tmp_path, world, root, empty_rank):
    """pco_gfx_gather_chunks / pco_gfx_scatter_chunks with rank != root, a root that is not rank 0 and a rank that owns no bytes, executed
    by `world` PROCESSES on ONE device: RCCL itself refuses duplicate devices, so the nine nccl* calls are served by the loopback
    transport of tests/fake_rccl.cpp (PCO_GFX_RCCL_LIB); everything above them -- the size exchange, the offset arithmetic, which rank
    posts what -- is the product's code, and the assembled body must be the oracle's chunks in chunk order."""
    import oracle_lib as O
    from pcodec_amd import _lib as G
    if G.lib().pco_gfx_device_count() < 1:
        pytest.skip("needs a HIP device")
    outs = _run_ranks(tmp_path, world, f"root={root}, one_device=True, empty_rank={empty_rank}",
                      {"PCO_GFX_RCCL_LIB": _fake_rccl_lib(), "PCO_FAKE_RCCL_DIR": str(tmp_path)})
    kw = dict(mode=1, delta=2, delta_order=1)
    want = b"".join(U_chunk(O, a, kw) for a in _sharded_inputs(11))
    assert all(o[:2] == b"OK" for o in outs)
    assert outs[root][2:] == want
    assert all(len(o) == 2 for r, o in enumerate(outs) if r != root)


@pytest.mark.gpu
@pytest.mark.parametrize("world,root", [(2, 0), (3, 1)])
def test_c_abi_gather_scatter_fail_collectively(tmp_path, world, root):
    """A root without room / without a buffer, a receiver without room: every rank returns INVALID_ARGUMENT, nobody blocks in a send
    or receive whose partner threw, and the communicator is still usable (the advisor's round-3 finding on pco_gfx_comm.inc)."""
    from pcodec_amd import _lib as G
    if G.lib().pco_gfx_device_count() < 1:
        pytest.skip("needs a HIP device")
    outs = _run_ranks(tmp_path, world, f"root={root}, one_device=True, failure=True",
                      {"PCO_GFX_RCCL_LIB": _fake_rccl_lib(), "PCO_FAKE_RCCL_DIR": str(tmp_path)})
    assert all(o[:2] == b"OK" for o in outs), outs


@pytest.mark.gpu
def test_c_abi_gather_scatter_two_ranks_over_rccl(tmp_path):
    """Two processes, two devices, pco_gfx_gather_chunks / pco_gfx_scatter_chunks over RCCL (the unique id travels through a file, as a
    Rust host would pass it over its own channel).  Skipped on a single-GPU box."""
    import subprocess
    import oracle_lib as O
    from pcodec_amd import _lib as G
    if G.lib().pco_gfx_device_count() < 2:
        pytest.skip("needs two HIP devices")
    world = 2
    idp, outp = str(tmp_path / "id"), str(tmp_path / "out")
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", f"import sys; sys.path.insert(0, {HERE!r}); sys.path.insert(0, {os.path.join(HERE, '..')!r}); import test_sharding as T; T._c_abi_rank_main({r}, {world}, {idp!r}, {outp!r})"], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    kw = dict(mode=1, delta=2, delta_order=1)
    want = b"".join(U_chunk(O, a, kw) for a in _sharded_inputs(11))
    r0 = open(outp + ".0", "rb").read(); r1 = open(outp + ".1", "rb").read()
    assert r0[:2] == b"OK" and r1[:2] == b"OK"
    assert r0[2:] == want
