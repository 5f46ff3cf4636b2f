"""Test infrastructure: a ``BlockChannel`` transport on an initialised ``torch.distributed`` group (gloo on the CPU), so that the
multi-rank host logic of sdrterm_amd/parallel.py and sdrterm_amd/dsp/vfo_processor.py runs with world_size 2 on a box without a
GPU.  The product's transport is ``sdrterm_amd.parallel.RcclTransport`` (RCCL through the C ABI); nothing under sdrterm_amd/
imports torch."""
from typing import List, Optional


class TorchTransport:
    """Broadcast through an initialised ``torch.distributed`` group.  Blocks are host (gloo) or device (nccl) uint8 tensors."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device if device is not None else torch.device('cpu')
        self.on_device = self.device.type != 'cpu'

    def bcast_bytes(self, data: Optional[bytes], nbytes: int, root: int = 0) -> bytes:
        t = self._torch.empty(nbytes, dtype=self._torch.uint8, device=self.device)
        if self.rank == root:
            t.copy_(self._torch.frombuffer(bytearray(data), dtype=self._torch.uint8))
        self._dist.broadcast(t, src=root, group=self._group)
        return t.cpu().numpy().tobytes()

    def bcast_block(self, data: Optional[bytes], nbytes: int, root: int = 0, slot: int = 0):
        """('host', bytes): the block as every rank's engine takes it from host memory."""
        return 'host', self.bcast_bytes(data, nbytes, root)

    def block_consumed(self, slot: int, stream: int = 0) -> None:
        pass

    def allgather_bytes(self, data: bytes) -> List[bytes]:
        """Every rank's ``data`` (same length everywhere), in rank order."""
        mine = self._torch.frombuffer(bytearray(data), dtype=self._torch.uint8).to(self.device)
        parts = [self._torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(parts, mine, group=self._group)
        return [p.cpu().numpy().tobytes() for p in parts]

    def close(self):
        pass


def max_over_ranks(value: float, device=None) -> float:
    """Wall time of the slowest rank (bench.py contract)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
