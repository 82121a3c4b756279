#!/usr/bin/env python3
"""Profile-1 kernel probe (cfg 5 geometry: N = 2048 stereo s16, hop 1920): K7 / K8 time for a 60 s and a 10 min clip.
FRAD_TUNE_NO_WAVE_P1=1 selects the one-shot kernels for comparison."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core, _lib
if os.environ.get("FRAD_PROBE_LIB"):                     # diagnostic builds of libfrad_hip.so (kernel ablations)
    _lib.LIB_PATH = os.environ["FRAD_PROBE_LIB"]
dev = torch.device("cuda:0")
def timeit(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
N, C, hop = 2048, 2, 1920
g = torch.Generator(device=dev).manual_seed(1)
loss = 1.25 ** 20 / 19 + 0.5
for secs in (60, 600):
    n = secs * 48000
    F = (n - N) // hop + 1
    pcm = (torch.randn((n, C), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
    q, tq = core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)
    t7 = timeit(lambda: core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop))
    t8 = timeit(lambda: core.p1_digital_batch(q, tq, N, C, 16, 48000))
    b7 = F * (N * C * 2 + N * C * 4 + 27 * C * 4); b8 = F * (N * C * 4 + 27 * C * 4 + N * C * 8)
    print(json.dumps({"secs": secs, "frames": F, "wave": not os.environ.get("FRAD_TUNE_NO_WAVE_P1"),
                      "K7_ms": round(t7, 4), "K7_frac": round(b7 / t7 / 1e6 / 8000, 4),
                      "K8_ms": round(t8, 4), "K8_frac": round(b8 / t8 / 1e6 / 8000, 4)}))
