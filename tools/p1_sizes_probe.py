#!/usr/bin/env python3
"""Profile-1 K7 / K8 time per sample across compact frame sizes (stereo s16, ~58 M samples per launch): the {160, 192, 224} x 2^n
sizes (mixed-radix kernels, frad_mixed.hip; the widest ones through the HBM workspace path) next to the powers of two."""
import json, os, sys
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
C = 2
g = torch.Generator(device=dev).manual_seed(1)
loss = 1.25 ** 20 / 19 + 0.5
sizes = [int(x) for x in sys.argv[1:]] or [1024, 1280, 1536, 1792, 2048, 2560, 3584, 4096, 5120, 7168, 8192, 28672]
for N in sizes:
    F = max(1, (14999 * 2048) // N) if N <= 8192 else 64
    pcm = (torch.randn((F * N, C), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
    q, tq = core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss)
    t7 = timeit(lambda: core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss))
    t8 = timeit(lambda: core.p1_digital_batch(q, tq, N, C, 16, 48000))
    S = F * N * C
    print(json.dumps({"N": N, "frames": F, "K7_ms": round(t7, 4), "K8_ms": round(t8, 4), "K7_ns_per_sample": round(t7 * 1e6 / S, 4),
                      "K8_ns_per_sample": round(t8 * 1e6 / S, 4)}), flush=True)
