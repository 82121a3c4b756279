#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

Metric   : Msamples/s encode+decode, 48 kHz stereo, frame = 2048 (sample = one PCM value).
Workload : configs[1] -- 10 min of 48 kHz stereo s16le PCM, profile 0 (DCT archiving), 32-bit
           big-endian storage, N = 2048: 14 062 full frames + the 1 024-sample tail frame, batched
           on one MI355X.  Synthetic "signal A" (harmonic mix + -60 dBFS noise), generated on the GPU.
Step     : one pass of the hot path over the whole clip: analogue (encode) of every frame, then
           digital (decode) of every payload; inputs are resident in HBM when the clock starts.
N > 1    : frames are independent -> every rank owns one such clip (weak scaling), no collective
           on the data path; the only collectives are the timing barrier and the max-reduce.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel: algorithmic bytes per launch
(encode B_in + b/8 = 6 B/sample, decode b/8 + 8 = 12 B/sample) / its mean HIP-event time.
`cpu_baseline` times the oracle (NumPy/SciPy restatement of the reference path) on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from frad_python_amd import core  # noqa: E402
from frad_python_amd.parallel import Timer  # noqa: E402

SRATE, CHANNELS, FSIZE, BITS = 48000, 2, 2048, 32
SECONDS = 600
HBM_PEAK_GBS = 8000.0                     # MI355X HBM3E peak (MI355X_MICROARCH.md)


def signal_a_gpu(n: int, channels: int, srate: int, seed: int, device) -> torch.Tensor:
    """"Signal A" of SURVEY 8d as s16le, synthesised on the device: 8 partials of 110*(c+1) Hz with
    0.5 Hz AM, peak 0.8, plus -60 dBFS white noise."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64, device=device) / srate
    out = torch.empty((n, channels), dtype=torch.float64, device=device)
    for c in range(channels):
        ph = torch.rand(8, generator=g, device=device, dtype=torch.float64) * 2 * np.pi
        x = torch.zeros(n, dtype=torch.float64, device=device)
        for h in range(8):
            x += torch.sin(2 * np.pi * 110.0 * (c + 1) * (h + 1) * t + ph[h]) / (h + 1)
        x *= 0.75 + 0.25 * torch.sin(2 * np.pi * 0.5 * t + c)
        out[:, c] = x
    out *= 0.8 / out.abs().max()
    out += torch.randn((n, channels), generator=g, device=device, dtype=torch.float64) * 1e-3
    return torch.clamp(torch.round(out * 32768.0), -32768, 32767).to(torch.int16)


PREWARM = 120                                                 # untimed clock-ramp steps before the warm-up proper


class Workload:
    def __init__(self, device, seed):
        self.n_total = SRATE * SECONDS                        # sample-frames
        self.n_full = self.n_total // FSIZE                   # 14062
        self.tail = self.n_total - self.n_full * FSIZE        # 1024
        self.pcm = signal_a_gpu(self.n_total, CHANNELS, SRATE, seed, device)
        self.samples = self.n_total * CHANNELS
        lib = core._lib.load()
        lib.plan_prepare(FSIZE, False)
        if self.tail:
            lib.plan_prepare(self.tail, False)
        nb = lib.payload_bytes(FSIZE, CHANNELS, BITS)
        self.pay = torch.empty((self.n_full, nb), dtype=torch.uint8, device=device)
        self.absmax_all = torch.empty(self.n_full + 1, dtype=torch.float64, device=device)   # main frames + tail frame
        self.absmax = self.absmax_all[:self.n_full]
        self.out = torch.empty((self.n_full, FSIZE, CHANNELS), dtype=torch.float64, device=device)
        if self.tail:
            nbt = lib.payload_bytes(self.tail, CHANNELS, BITS)
            self.pay_t = torch.empty((1, nbt), dtype=torch.uint8, device=device)
            self.absmax_t = self.absmax_all[self.n_full:]
            self.out_t = torch.empty((1, self.tail, CHANNELS), dtype=torch.float64, device=device)
        self.tail_pcm = self.pcm[self.n_full * FSIZE:]
        self.over = torch.zeros((), dtype=torch.int32, device=device)
        # the clip's last, short frame is independent of the 14 062 full ones (own input slice, own buffers): it runs on
        # a second HIP stream so that its tiny launches overlap the big batch instead of queueing behind it.  The two
        # streams never wait for each other inside the timed region -- the closing torch.cuda.synchronize() covers both.
        self.side = torch.cuda.Stream(device=device)
        self.side.wait_stream(torch.cuda.current_stream())          # the synthetic clip is ready before the first tail

    def tail_frame(self):
        """encode + decode of the short last frame, on the side stream"""
        if not self.tail:
            return
        with torch.cuda.stream(self.side):
            core.analogue_batch(0, self.tail_pcm, "s16le", 1, self.tail, CHANNELS, BITS, False,
                                check_overflow=False, out=self.pay_t, absmax=self.absmax_t)
            core.overflow_scan(self.absmax_t, BITS, self.over)
            core.digital_batch(0, self.pay_t, 1, self.tail, CHANNELS, BITS, False, out=self.out_t)

    def encode(self, ev=None):
        if ev: ev[0].record()
        core.analogue_batch(0, self.pcm, "s16le", self.n_full, FSIZE, CHANNELS, BITS, False,
                            check_overflow=False, out=self.pay, absmax=self.absmax)
        if ev: ev[1].record()
        self.tail_frame()            # side stream: fills CUs as the resident kernels' blocks retire

    def decode(self, ev=None):
        if ev: ev[0].record()
        core.digital_batch(0, self.pay, self.n_full, FSIZE, CHANNELS, BITS, False, out=self.out)
        if ev: ev[1].record()

    def overflow_check(self):
        """The reference's per-frame overflow test (profile0.py:24-26): evaluated on the device every
        step over all frames (one launch, frad_p0_overflow_scan; the tail frame's on its own stream); the
        host reads the sticky flag once, after the timed region."""
        core.overflow_scan(self.absmax, BITS, self.over)


def cpu_baseline(pcm_host: np.ndarray, n_frames: int, min_seconds: float = 12.0):
    """Oracle, one frame at a time on one core -- the reference's own cost profile (encoder.py:60,
    decoder.py:55 loop one frame per iteration through numpy/scipy)."""
    from oracle import frad_oracle as fo
    dt = fo.pcm_dtype("s16le")
    t0, done = time.perf_counter(), 0
    while time.perf_counter() - t0 < min_seconds:             # whole passes over the sample, >= min_seconds of work
        for f in range(n_frames):
            frame = fo.to_f64(pcm_host[f * FSIZE:(f + 1) * FSIZE], dt)
            frad, idx, ch, sr = fo.p0_analogue(frame, BITS, SRATE, False)
            fo.p0_digital(frad, idx, ch, False)
        done += n_frames
    dt_s = time.perf_counter() - t0
    return done * FSIZE * CHANNELS / dt_s / 1e6, dt_s, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~0.33 ms of GPU work and the card needs ~10 ms of sustained load to reach its steady clocks
    # (per-step time falls from 0.39 to 0.325 ms over the first ~25 steps, FRAD_BENCH_TRACE=1 shows it):
    # the defaults warm up past that ramp and time long enough for a stable figure (still < 0.2 s of GPU time)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    # one process per GPU; FRAD_BENCH_BACKEND=gloo lets several ranks rehearse on fewer GPUs (ranks then share
    # devices round-robin) -- the data path has no collective either way
    backend = os.environ.get("FRAD_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = Workload(dev, seed=1234 + rank)

    # untimed: whatever --warmup says, run at least PREWARM steps (~40 ms) first so that the card is at its steady
    # clocks when the W warm-up steps and the K timed steps run (the ramp is ~10 ms, see the defaults above)
    for _ in range(max(0, PREWARM - args.warmup)):
        wl.encode(); wl.decode(); wl.overflow_check()
    for _ in range(args.warmup):
        wl.encode(); wl.decode(); wl.overflow_check()
    ev_enc = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_dec = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def timed_steps():
        for i in range(args.steps):
            wl.encode(ev_enc[i])
            wl.decode(ev_dec[i])
            wl.overflow_check()
    # barrier + torch.cuda.synchronize() on both sides of exactly K steps, MAX over ranks (parallel.Timer)
    elapsed = Timer(dist, torch.cuda.synchronize).measure(timed_steps)
    assert int(wl.over.item()) == 0, "synthetic audio must not overflow float32 storage"

    if os.environ.get("FRAD_BENCH_TRACE"):                   # per-step start-to-start times, for diagnosis only
        print("step starts (ms):", [round(ev_enc[i][0].elapsed_time(ev_enc[i + 1][0]), 3) for i in range(args.steps - 1)],
              "wall", round(elapsed * 1e3, 3), file=sys.stderr)
    enc_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_enc]))
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_dec]))
    full = wl.n_full * FSIZE * CHANNELS                      # samples one main launch processes
    enc_bytes, dec_bytes = full * (2 + BITS // 8), full * (BITS // 8 + 8)

    def roof(name, nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": nbytes,
                "avg_launch_ms": round(ms, 4)}
    r_enc = roof("k_p0_fwd_unit<double,PlanA10,s16,C=2> (load+to_f64+DCT-II+absmax+f32 cast+BE pack)", enc_bytes, enc_ms)
    r_dec = roof("k_p0_inv_unit<PlanA10,32,C=2> (unpack+scrub+inverse DCT+f64 interleaved store)", dec_bytes, dec_ms)
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this round (profiles/, collected with
    # tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE x2 as the gfx950 guide prescribes)
    try:
        cnt = json.load(open(os.path.join(ROOT, "profiles", "r01_counters.json")))
        for r, key in ((r_enc, "k_p0_fwd"), (r_dec, "k_p0_inv")):
            hit = [v for k, v in cnt.items() if key in k and v.get("launch", {}).get("Grid_Size") and "hbm_bytes_per_launch_corrected" in v
                   and int(v["hbm_bytes_per_launch_corrected"]) > 1e8]
            if hit:
                r["traffic"] = int(hit[0]["hbm_bytes_per_launch_corrected"])
    except Exception:
        pass
    dominant, other = (r_dec, r_enc) if dec_ms >= enc_ms else (r_enc, r_dec)

    if rank == 0:
        value = world * wl.samples * args.steps / elapsed / 1e6
        line = {
            "metric": "Msamples/s encode+decode, 48 kHz stereo frame=2048; achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: 10 min 48 kHz stereo s16le, profile 0, bits=32 BE, frame=2048, "
                                   "14062 frames + 1024-sample tail per GPU, encode then decode",
                       "prewarm_steps": max(0, PREWARM - args.warmup), "frames_per_gpu": wl.n_full + (1 if wl.tail else 0), "samples_per_gpu": wl.samples,
                       "parallelism": f"frames sharded over {world} GPU(s), no collectives"},
            "roofline": dominant, "roofline_other": other,
            "hbm_frac_enc_plus_dec": round((enc_bytes + dec_bytes) / ((enc_ms + dec_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            n = wl.n_full
            host = wl.pcm[:n * FSIZE].cpu().numpy()
            v, secs, done = cpu_baseline(host, n)
            line["cpu_baseline"] = {"value": round(v, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                    "sample": f"the clip's {n} full frames, {done // n} pass(es) = {done * FSIZE * CHANNELS} samples "
                                              f"encode+decode, one frame per call like the reference loop, "
                                              f"{secs:.1f} s of CPU work",
                                    "host_cpus": os.cpu_count()}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
