"""Multi-GPU layout of the hot path: independent frames (or trials) are block-partitioned over the
ranks - one process per GPU - each rank computes the TDOA table of its own frames with no data-path
exchange, and ONE all-gather of the fixed-size record tables assembles the job's table on every
rank.  The reference is single-process; there is nothing to translate.

Two gather transports:
  * ``gather_tables_rccl`` - the engine's own RCCL communicator (ncclAllGather over xGMI on the
    engine's HIP stream, device buffers, no host hop) - the product path on a GPU node;
  * ``gather_tables_torch`` - ``torch.distributed.all_gather`` on host tensors (gloo) - used by the
    CPU tests of the sharding logic and as the reported fallback when RCCL cannot initialise.
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import numpy as np

from ._ffi import RECORD


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ``total`` units owned by ``rank``: sizes differ by at most one, lower ranks
    get the larger blocks."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int) -> list:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def gather_tables_torch(local: np.ndarray, total_frames: int, rank: int, world: int) -> np.ndarray:
    """All-gather of per-rank tables [frames_r][P] (structured RECORD rows) through torch.distributed;
    shards may differ in size by one frame, so every rank pads to the largest shard."""
    import torch
    import torch.distributed as dist
    sizes = shard_sizes(total_frames, world)
    width = local.shape[1]
    padded = np.zeros((max(sizes), width), dtype=RECORD)
    padded[: local.shape[0]] = local
    send = torch.from_numpy(padded.view(np.uint8).reshape(-1).copy())
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    parts = [r.numpy().view(RECORD).reshape(max(sizes), width)[: sizes[k]] for k, r in enumerate(recv)]
    return np.concatenate(parts, axis=0)


def gather_blocks_torch(local: np.ndarray, sizes: Sequence[int]) -> np.ndarray:
    """All-gather of per-rank record blocks local[sizes[rank]] (1-D, structured RECORD rows) of unequal size: every rank
    pads to the largest block; the concatenation in rank order is returned on every rank."""
    import torch
    import torch.distributed as dist
    world = len(sizes)
    padded = np.zeros(max(sizes), dtype=RECORD)
    padded[: local.shape[0]] = local
    send = torch.from_numpy(padded.view(np.uint8).reshape(-1).copy())
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    return np.concatenate([r.numpy().view(RECORD)[: sizes[k]] for k, r in enumerate(recv)])


def shard_pairs(mics: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of the ordered pair list (row-major i<j, main.py:202-203) owned by ``rank`` when ONE
    frame is split over the GPUs (configuration C4: 256 microphones, 32 640 pairs -> 4080 per GPU on 8)."""
    return shard_range(mics * (mics - 1) // 2, rank, world)


def sharded_pair_table(frame: np.ndarray, rank: int, world: int, compute_block: Callable[[np.ndarray, np.ndarray], np.ndarray],
                       gather=gather_blocks_torch) -> np.ndarray:
    """Single large frame[M][L]: every rank holds the whole frame (the spectra are recomputed locally - cheaper than
    moving them: 256 forward transforms against 4080 pair transforms per rank), computes the rows of its own block of the
    pair list with ``compute_block(frame, pairs[start:stop])`` and ONE gather assembles table[P] on every rank."""
    from .engine import pair_list
    m = frame.shape[0]
    pairs = pair_list(m)
    sizes = [shard_pairs(m, r, world)[1] - shard_pairs(m, r, world)[0] for r in range(world)]
    lo, hi = shard_pairs(m, rank, world)
    if min(sizes) <= 0:                                     # decided identically on EVERY rank, before any collective (no rank waits for one that raised)
        raise ValueError("more ranks than pairs")
    local = compute_block(frame, pairs[lo:hi])
    if local.shape != (hi - lo,):
        raise ValueError("compute_block must return one record per pair of the block")
    return gather(local, sizes)


def gather_tables_rccl(engine, d_local: int, d_all: int, frames_per_rank: int, pairs: int) -> None:
    """One ncclAllGather of equal-size device tables (frames_per_rank * pairs records per rank)."""
    engine.all_gather_dev(d_local, d_all, frames_per_rank * pairs * RECORD.itemsize)


def sharded_tdoa(frames_of: Callable[[int, int], np.ndarray], total_frames: int, rank: int, world: int,
                 compute: Callable[[np.ndarray], np.ndarray], gather=gather_tables_torch) -> np.ndarray:
    """frames_of(first, count) builds this rank's frames, ``compute`` maps frames[B][M][L] to a table[B][P];
    returns the whole job's table[total_frames][P] on every rank."""
    lo, hi = shard_range(total_frames, rank, world)
    if total_frames < world:                                 # decided identically on every rank, before any collective
        raise ValueError("more ranks than frames")
    local = compute(frames_of(lo, hi - lo))
    return gather(local, total_frames, rank, world)
