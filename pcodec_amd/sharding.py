"""Chunk sharding across ranks.

Chunks are independent (each carries its own ChunkMeta and page, standalone/simple.rs:62-91), so a many-chunk
input is partitioned into contiguous blocks of chunk indices, one block per GPU / rank; every rank encodes or
decodes its block with no collective on the data path.  Only when one rank needs the whole `.pco` file are the
compressed chunks gathered:

  1. every rank compacts its chunks into one contiguous byte range on the device (pco_gfx_compact_chunks);
  2. an all-gather of (chunk count, byte count) per rank -- 16 bytes each -- and of the per-chunk sizes;
  3. a gather-v of the byte ranges: the root posts one receive per peer straight into its file buffer at that peer's
     byte offset, every peer sends exactly its bytes (one grouped batch of point-to-point operations: over RCCL that is
     one ncclGroup of ncclSend / ncclRecv, each pair on its own xGMI link; nothing is padded to the largest rank).

The decode direction is the mirror image: the root scatters byte ranges, every rank decodes its block.
With the `nccl` backend this runs over RCCL / xGMI; the `gloo` backend is what the CPU tests use.
torch.distributed is plumbing here; nothing in this module touches the codec.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_chunks: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the chunk indices owned by `rank`: chunk c lives on rank floor(c * world / n_chunks)'s block,
    i.e. contiguous blocks whose sizes differ by at most one and whose concatenation is the original order."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, rem = divmod(n_chunks, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_of_chunk(c: int, n_chunks: int, world: int) -> int:
    """Inverse of shard_range."""
    base, rem = divmod(n_chunks, world)
    split = rem * (base + 1)
    if c < split:
        return c // (base + 1)
    return rem + (c - split) // base


def pack_chunks(chunks: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate a rank's compressed chunks on the host: (uint8 payload, int64 sizes).  (Device-resident chunks are
    compacted by pco_gfx_compact_chunks instead.)"""
    sizes = np.array([len(c) for c in chunks], dtype=np.int64)
    payload = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if len(chunks) else np.zeros(0, np.uint8)
    return payload, sizes


def gather_sizes(sizes, group=None):
    """All-gather of the per-chunk compressed sizes.  `sizes` is a 1-D int64 torch tensor (any device the backend
    supports); blocks may differ in length by one chunk, so they are padded to the longest block with -1."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = torch.tensor([sizes.numel()], dtype=torch.int64, device=sizes.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    longest = int(max(int(c.item()) for c in counts))
    padded = torch.full((longest,), -1, dtype=torch.int64, device=sizes.device)
    padded[: sizes.numel()] = sizes
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[: int(c.item())] for o, c in zip(out, counts)]


def exchange_totals(n_bytes: int, device, group=None) -> List[int]:
    """All-gather of every rank's compacted byte count (8 bytes per rank)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = torch.tensor([int(n_bytes)], dtype=torch.int64, device=device)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    return [int(g.item()) for g in got]


def _p2p(ops):
    import torch.distributed as dist
    if not ops:
        return
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def gather_stream(payload, n_bytes: int, dst: int = 0, group=None, out=None, out_offset: int = 0, totals: Optional[List[int]] = None):
    """Gather every rank's compacted chunk bytes on rank `dst`, in rank (= chunk) order, at exact sizes.

    payload: 1-D uint8 torch tensor whose first n_bytes bytes are this rank's chunks back to back.
    out (on dst): 1-D uint8 tensor receiving the stream from byte out_offset on (allocated if None).
    Returns (out or None, per-rank byte offsets relative to out_offset with the total appended)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if totals is None:
        totals = exchange_totals(n_bytes, payload.device, group)
    offs = [0]
    for t in totals:
        offs.append(offs[-1] + t)
    if rank == dst:
        if out is None:
            out = torch.empty(out_offset + offs[-1] + 64, dtype=torch.uint8, device=payload.device)
        if out.numel() < out_offset + offs[-1]:
            raise ValueError("gather_stream: destination too small")
        ops = [dist.P2POp(dist.irecv, out[out_offset + offs[r]: out_offset + offs[r + 1]], r, group)
               for r in range(world) if r != dst and totals[r] > 0]
        out[out_offset + offs[dst]: out_offset + offs[dst + 1]].copy_(payload[:n_bytes])
        _p2p(ops)
        return out, offs
    if n_bytes > 0:
        _p2p([dist.P2POp(dist.isend, payload[:n_bytes], dst, group)])
    return None, offs


def scatter_stream(stream, offs: Sequence[int], recv, src: int = 0, group=None, stream_offset: int = 0):
    """The decode direction: rank `src` holds the chunk stream; rank r receives bytes [offs[r], offs[r+1]) into `recv`
    (1-D uint8 tensor of at least that size + the decoder's 16 bytes of slack).  Returns the number of bytes received."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = offs[rank + 1] - offs[rank]
    if recv.numel() < mine:
        raise ValueError("scatter_stream: receive buffer too small")
    if rank == src:
        ops = [dist.P2POp(dist.isend, stream[stream_offset + offs[r]: stream_offset + offs[r + 1]], r, group)
               for r in range(world) if r != src and offs[r + 1] > offs[r]]
        recv[:mine].copy_(stream[stream_offset + offs[src]: stream_offset + offs[src + 1]])
        _p2p(ops)
    elif mine > 0:
        _p2p([dist.P2POp(dist.irecv, recv[:mine], src, group)])
    return mine


def gather_pages(payload, sizes, dst: int = 0, group=None):
    """Convenience form used by the tests: gather sizes and bytes; returns on `dst` (list of per-rank uint8 tensors,
    list of per-rank size tensors), elsewhere (None, sizes)."""
    all_sizes = gather_sizes(sizes, group)
    totals = [int(s.sum().item()) for s in all_sizes]
    import torch.distributed as dist
    out, offs = gather_stream(payload, int(sizes.sum().item()), dst, group, totals=totals)
    if dist.get_rank(group) == dst:
        return [out[offs[r]: offs[r + 1]] for r in range(len(totals))], all_sizes
    return None, all_sizes


def assemble_standalone_file(header: bytes, per_rank_payloads: Sequence[bytes]) -> bytes:
    """header (pco_gfx_write_standalone_header) + every rank's chunks in rank order + the 0x00 terminator
    (standalone/constants.rs:5)."""
    return header + b"".join(per_rank_payloads) + b"\x00"


def split_payload(payload: bytes, sizes: Sequence[int]) -> List[bytes]:
    out = []; pos = 0
    for s in sizes:
        out.append(payload[pos: pos + int(s)]); pos += int(s)
    return out


# ------------------------------------------------------------------------------------------------ the same over the C ABI
class Comm:
    """include/pco_gfx.h section 5: the gather-v / scatter of chunk bytes through libpco_gfx.so itself (RCCL called directly, no
    torch.distributed) -- what a Rust / C host binds.  Rank 0 makes the 128-byte id (Comm.unique_id()) and hands it to the other
    ranks over its own channel; every rank then builds Comm(id, world, rank) on its device."""

    def __init__(self, ident: bytes, world: int, rank: int):
        import ctypes as C
        from . import _lib as G
        self._G, self._C, self._L = G, C, G.lib()
        L = self._L
        L.pco_gfx_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pco_gfx_comm_free.argtypes = [C.c_void_p]
        L.pco_gfx_gather_chunks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.pco_gfx_scatter_chunks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        self.world, self.rank = world, rank
        self._h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(ident))
        G.check(L.pco_gfx_comm_init(buf, world, rank, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib as G
        L = G.lib()
        L.pco_gfx_comm_unique_id.argtypes = [C.c_void_p]
        buf = (C.c_ubyte * 128)()
        G.check(L.pco_gfx_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pco_gfx_comm_free(self._h); self._h = None

    __del__ = close

    def gather(self, stream_ptr: int, n_bytes: int, file_ptr: int = 0, file_cap: int = 0, file_offset: int = 0, root: int = 0, hip_stream=None) -> List[int]:
        """pco_gfx_gather_chunks on raw device pointers; returns the per-rank byte offsets (total last), on every rank."""
        C = self._C
        offs = (C.c_uint64 * (self.world + 1))()
        self._G.check(self._L.pco_gfx_gather_chunks(self._h, root, stream_ptr, n_bytes, file_ptr, file_cap, file_offset, offs, hip_stream))
        return [int(x) for x in offs]

    def scatter(self, file_ptr: int, offsets: Sequence[int], recv_ptr: int, recv_cap: int, file_offset: int = 0, root: int = 0, hip_stream=None) -> int:
        """pco_gfx_scatter_chunks; returns this rank's byte count."""
        C = self._C
        offs = (C.c_uint64 * (self.world + 1))(*[int(x) for x in offsets])
        got = C.c_uint64(0)
        self._G.check(self._L.pco_gfx_scatter_chunks(self._h, root, file_ptr, file_offset, offs, recv_ptr, recv_cap, C.byref(got), hip_stream))
        return int(got.value)
