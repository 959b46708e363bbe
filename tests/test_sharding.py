"""N > 1 path on CPU: world-size-2 gloo run of the chunk sharding + page gather.
The per-chunk codec here is the oracle (test infrastructure); on the GPU box bench.py --gather runs the same
sharding code over RCCL with libpco_gfx as the codec."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
