"""N > 1 path on CPU: two gloo ranks each encode their contiguous frame range; concatenated in rank
order the payloads equal the single-rank result bit for bit, and the timing reduction takes the max."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frad_python_amd import synth
from frad_python_amd.parallel import Timer, overlapped_shard, shard_range


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 14063, 4096):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    a, b, s0, s1 = overlapped_shard(10, 1920, 2048, 1, 2)
    assert (a, b, s0, s1) == (5, 10, 5 * 1920, 9 * 1920 + 2048)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    from helpers import OracleBridge
    br = OracleBridge()
    F, N, C = 11, 256, 2
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, 48000, seed=4), "s16le")
    a, b = shard_range(F, rank, world)
    timer = Timer(dist)
    res = {}

    def work():
        res["frames"] = br.lossless_encode(0, raw[a * N:b * N].tobytes(), "s16le", b - a, N, C, 32, False)
        if rank == 1:
            import time
            time.sleep(0.3)
    dt = timer.measure(work)
    gathered = [None] * world
    dist.all_gather_object(gathered, [hashlib.sha256(f[0]).hexdigest() for f in res["frames"]])   # verification only
    if rank == 0:
        whole = br.lossless_encode(0, raw.tobytes(), "s16le", F, N, C, 32, False)
        q.put(([h for part in gathered for h in part] == [hashlib.sha256(f[0]).hexdigest() for f in whole], dt))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    ok, dt = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert ok
    assert dt >= 0.3            # MAX over ranks: rank 1 slept


def test_bench_launcher_starts_one_process_per_gpu():
    """`bench.py --gpus 2` without a torch.distributed environment: the parent starts two ranks (gloo rehearsal on CPU,
    --dry-run: rendezvous, barrier and MAX-reduce only) and rank 0 prints the one JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["clips_rank0"] == 2048
    assert d["ms_per_step"] >= 20.0                          # MAX over ranks: rank 1 sleeps 20 ms


def test_bench_launcher_ends_the_job_when_a_rank_is_lost():
    """A rank that dies before the rendezvous (no such GPU, out of memory) must not leave the others waiting for ever: the launcher
    stops them and returns the failure (seen on a one-GPU box: `bench.py --gpus 2` sat silent until the box's watchdog)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FRAD_BENCH_DRYRUN_DIE"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 100
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
