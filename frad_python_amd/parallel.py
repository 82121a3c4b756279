"""Multi-GPU layout of the transform core: frames (and clips) are independent units, so a batch is
split into contiguous per-rank ranges and every rank runs the single-GPU path on its own range --
no collective on the data path.  One process per GPU (``torch.distributed``; the
``nccl`` backend is RCCL on ROCm); the only collectives are the timing barrier and the max-reduce of
the measured interval used by bench.py, exercised on CPU with ``gloo`` in tests/test_parallel.py."""
from __future__ import annotations

import time


def shard_range(n_units: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range [start, stop) of rank's units: floor(n*r/world) .. floor(n*(r+1)/world)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return n_units * rank // world, n_units * (rank + 1) // world


def overlapped_shard(n_frames: int, hop: int, n: int, rank: int, world: int) -> tuple[int, int, int, int]:
    """Compact profiles: rank's frame range and the sample-frame range of the source it must read.
    Frame i covers sample-frames [i*hop, i*hop + n): shards only overlap in what they READ (the halo)."""
    a, b = shard_range(n_frames, rank, world)
    return a, b, a * hop, (b - 1) * hop + n if b > a else a * hop


class Timer:
    """bench.py's timed region: barrier + device sync on both sides, MAX over ranks."""

    def __init__(self, dist=None, device_sync=None):
        self.dist, self.sync = dist, device_sync or (lambda: None)

    def fence(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
        self.sync()

    def measure(self, fn) -> float:
        """Seconds for fn() on the slowest rank.  `self.per_rank` afterwards holds every rank's own time to its local device
        sync (before the closing barrier), so that a straggler shows up in rank 0's report."""
        import torch
        self.fence()
        t0 = time.perf_counter()
        fn()
        self.sync()
        local = time.perf_counter() - t0                      # this rank's work, not yet waiting for the others
        self.fence()
        dt = time.perf_counter() - t0
        self.per_rank = [local]
        if self.dist is not None and self.dist.is_initialized():
            nccl = self.dist.get_backend() == "nccl"
            t = torch.tensor([dt], dtype=torch.float64)
            mine = torch.tensor([local], dtype=torch.float64)
            if nccl:
                t, mine = t.cuda(), mine.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            every = [torch.zeros_like(mine) for _ in range(self.dist.get_world_size())]
            self.dist.all_gather(every, mine)
            dt = float(t.item())
            self.per_rank = [float(x.item()) for x in every]
        return dt
