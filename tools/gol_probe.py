#!/usr/bin/env python3
"""Exp-Golomb decode kernel probe: time vs batch size, with the clocks warmed by a streaming kernel first."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core
dev = torch.device("cuda:0")
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
N, C = 2048, 2
g = torch.Generator(device=dev).manual_seed(1)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
# two kinds of frames: noise-like (every bin busy) and tonal (a few hundred busy bins, the upper bands all zero: long runs of one
# code, which is where a speculative decoder has to work for its synchronisation)
for kind, F in (("noise", 64), ("noise", 1500), ("noise", 15000), ("tonal", 1500), ("tonal", 15000)):
    q = (torch.randn((F, N, C), generator=g, device=dev) * 6).round().to(torch.int32)
    if kind == "tonal":
        q = (torch.randn((F, N, C), generator=g, device=dev) * 40).round().to(torch.int32)
        q[:, 300:, :] = 0
    tq = torch.randint(0, 30, (F, 27, C), generator=g, device=dev, dtype=torch.int32)
    flat, offs = core.p1_golomb_encode_batch(q, tq)
    def dec():
        return core.p1_golomb_decode_batch(flat, offs, N, C)
    cold = timeit(dec)
    def warm_dec():
        for _ in range(20): big2.copy_(big)
        return dec()
    for _ in range(3): warm_dec()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): big2.copy_(big)
    a.record(); dec(); b.record(); torch.cuda.synchronize()
    dq, dt, st = dec()
    print(json.dumps({"kind": kind, "frames": F, "decode_ms": round(cold, 3), "decode_ms_after_20_copies": round(a.elapsed_time(b), 3),
                      "encode_ms": round(timeit(lambda: core.p1_golomb_encode_batch(q, tq)), 3),
                      "bytes_per_frame": int(offs[-1].item()) // F, "ok": bool(torch.equal(dq, q) and torch.equal(dt, tq))}))
