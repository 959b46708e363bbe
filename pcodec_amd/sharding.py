"""Chunk sharding across ranks.

Chunks are independent (each carries its own ChunkMeta and page, standalone/simple.rs:62-91), so a many-chunk
input is partitioned into contiguous blocks of chunk indices, one block per GPU / rank; every rank encodes or
decodes its block with no collective on the data path.  Only when one rank needs the whole `.pco` file are the
compressed chunks gathered: an all-gather of the per-chunk byte counts (8 B per chunk) followed by a gather of
the compressed bytes themselves.  With the `nccl` backend both run over RCCL / xGMI; the `gloo` backend is what
the CPU tests use.

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
    """Concatenate a rank's compressed chunks: (uint8 payload, int64 sizes)."""
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


def gather_pages(payload, sizes, dst: int = 0, group=None):
    """Gather every rank's compressed chunks on rank `dst`, in chunk order.

    payload: 1-D uint8 torch tensor holding this rank's compressed chunks back to back; sizes: their lengths.
    Returns on `dst`: (list of per-rank uint8 tensors trimmed to their true length, list of per-rank size tensors);
    on the other ranks: (None, all sizes).  The payload gather is padded to the largest per-rank total (the
    collective needs equal shapes); that is at most the compressed size of one rank's block.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    all_sizes = gather_sizes(sizes, group)
    totals = [int(s.sum().item()) for s in all_sizes]
    longest = max(totals) if totals else 0
    padded = torch.zeros(max(longest, 1), dtype=torch.uint8, device=payload.device)
    padded[: payload.numel()] = payload
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst, group=group)
        return [b[:t] for b, t in zip(bufs, totals)], all_sizes
    dist.gather(padded, None, dst=dst, group=group)
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
