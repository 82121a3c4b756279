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
        import torch
        self.fence()
        t0 = time.perf_counter()
        fn()
        self.fence()
        dt = time.perf_counter() - t0
        if self.dist is not None and self.dist.is_initialized():
            t = torch.tensor([dt], dtype=torch.float64)
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt
