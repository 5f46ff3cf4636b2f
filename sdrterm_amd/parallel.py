"""Multi-GPU partitioning of the hot path: one process per GPU, no data-path reduction.

* single VFO: chunks are independent (dsp_processor.py:140-149 keeps no state between them), so a stream of
  chunks is split into contiguous ranges, one per rank, with no collective at all;
* ``--simo``: rows (VFO channels) are independent given the raw block, so rows are split contiguously across
  ranks and the raw IQ block is broadcast from the rank that holds it (the only collective; RCCL over xGMI when
  the process group's backend is "nccl", gloo in the CPU tests).

These helpers are pure index arithmetic plus one ``torch.distributed.broadcast``; torch is imported lazily so the
single-GPU product path (ctypes only) never needs it.
"""
from typing import List, Tuple


def split_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of ``total`` items for ``rank`` of ``world`` (first ranks get the extras)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad world/rank')
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_rows(row_freqs: List[float], world: int, rank: int) -> Tuple[List[float], Tuple[int, int]]:
    """Rows (VFO channels) this rank computes, and their [start, stop) in the reference's row order."""
    lo, hi = split_range(len(row_freqs), world, rank)
    return list(row_freqs[lo:hi]), (lo, hi)


def shard_chunks(n_chunks: int, world: int, rank: int) -> Tuple[int, int]:
    return split_range(n_chunks, world, rank)


def broadcast_block(tensor, src: int = 0, group=None):
    """Broadcast the raw IQ block (a uint8/int16 tensor, device or host) from ``src`` to every rank, in place."""
    import torch
    import torch.distributed as dist
    # the block is opaque bytes to the collective: a uint8 view works for every backend (gloo has no int16)
    dist.broadcast(tensor.contiguous().view(torch.uint8) if tensor.dtype != torch.uint8 else tensor, src=src, group=group)
    return tensor


def max_over_ranks(value: float, device=None) -> float:
    """Wall time of the slowest rank (bench.py contract)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
