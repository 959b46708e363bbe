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


def _rank_round_trip(L, comm, rank, world, chunks_of_rank, cfg_kw):
    """One rank's part of the C-ABI file assembly: encode my block of chunks, compact, pco_gfx_gather_chunks to rank 0,
    pco_gfx_scatter_chunks back, decode.  Returns (file body on rank 0 else None, my decoded arrays)."""
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
    file_body = torch.zeros(file_cap if rank == 0 else 16, dtype=torch.uint8, device="cuda")
    G.check(L.pco_gfx_gather_chunks(comm, 0, payload.data_ptr(), total.value, file_body.data_ptr(), file_cap, 0, offsets, None))
    torch.cuda.synchronize()
    assert offsets[rank + 1] - offsets[rank] == total.value
    recv = torch.zeros(stream_cap, dtype=torch.uint8, device="cuda")
    got = C.c_uint64(0)
    G.check(L.pco_gfx_scatter_chunks(comm, 0, file_body.data_ptr(), 0, offsets, recv.data_ptr(), stream_cap - 16, C.byref(got), None))
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
    body = bytes(file_body[: offsets[world]].cpu().numpy()) if rank == 0 else None
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


def _c_abi_rank_main(rank, world, id_path, out_path):
    import torch
    from pcodec_amd import _lib as G
    torch.cuda.set_device(rank)
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
    from pcodec_amd.sharding import shard_range
    c0, c1 = shard_range(len(arrays), rank, world)
    kw = dict(mode=1, delta=2, delta_order=1)
    body, back = _rank_round_trip(L, comm, rank, world, arrays[c0:c1], kw)
    ok = all(np.array_equal(a, b) for a, b in zip(arrays[c0:c1], back))
    L.pco_gfx_comm_free(comm)
    with open(f"{out_path}.{rank}", "wb") as f:
        f.write(b"OK" if ok else b"NO")
        if body is not None:
            f.write(body)


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
