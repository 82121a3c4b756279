#!/usr/bin/env python3
"""Profile-1 kernel probe (cfg 5 geometry: N = 2048 stereo s16, hop 1920): K7 / K8 time for a 60 s and a 10 min clip.
FRAD_TUNE_NO_WAVE_P1=1 selects the one-shot kernels for comparison.

Last line: the UNFUSED yardstick of K7, as a lower bound built from parts that exist: (A) the profile-0 wave
encode of the same frames at 64-bit little-endian storage -- the same DCT, its coefficients to an HBM plane --, measured, plus
(B) a streaming quantiser priced at nothing but its bytes (coefficient plane in, q + tq out) moved at the speed of the
hand-written copy kernel, measured here on the same byte count.  A real (B) also has K7's 2 500 instructions per frame."""
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
    if secs == 600:
        # the same on a tonal signal (eight partials + a little noise): few busy bins with large codes, the upper bands quantised to
        # zero -- K7's exact fall-backs and K8's table misses depend on the data, the line above is noise
        tt = torch.arange(n, device=dev, dtype=torch.float64)
        sig = sum(a * torch.sin(2 * np.pi * f0 / 48000 * tt + ph) for a, f0, ph in
                  ((9000, 220.0, 0.1), (5000, 440.0, 1.0), (4000, 660.0, 2.0), (2500, 880.0, 0.5), (2000, 1320.0, 0.3), (1200, 2640.0, 1.7), (800, 5280.0, 2.9), (300, 9000.0, 0.7)))
        tone = (sig[:, None] * torch.tensor([1.0, 0.8], device=dev, dtype=torch.float64) + torch.randn((n, C), generator=g, device=dev) * 20).clamp(-32768, 32767).to(torch.int16)
        q2, tq2 = core.p1_analogue_batch(tone, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)
        t7t = timeit(lambda: core.p1_analogue_batch(tone, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop))
        t8t = timeit(lambda: core.p1_digital_batch(q2, tq2, N, C, 16, 48000))
        print(json.dumps({"secs": secs, "frames": F, "signal": "tonal", "K7_ms": round(t7t, 4), "K7_frac": round(b7 / t7t / 1e6 / 8000, 4),
                          "K8_ms": round(t8t, 4), "K8_frac": round(b8 / t8t / 1e6 / 8000, 4), "q_absmax": int(q2.abs().max().item()),
                          "q_zero_frac": round(float((q2 == 0).float().mean().item()), 3)}))
        del tt, sig, tone, q2, tq2
        out = torch.empty((F, N * C * 8), dtype=torch.uint8, device=dev)
        am = torch.empty(F, dtype=torch.float64, device=dev)
        ta = timeit(lambda: core.analogue_batch(0, pcm, "s16le", F, N, C, 64, True, frame_stride=hop, check_overflow=False, out=out, absmax=am))
        lib = _lib.load()
        nb = F * (N * C * 8 + N * C * 4 + 27 * C * 4) // 2 // 16 * 16          # a copy moves every byte twice (read + write)
        src = torch.empty(nb, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
        st = int(torch.cuda.current_stream().cuda_stream)
        tb = timeit(lambda: lib.bench_copy(src.data_ptr(), dst.data_ptr(), nb, st))
        print(json.dumps({"secs": secs, "frames": F, "K7_unfused_lower_bound_ms": round(ta + tb, 4), "A_p0_encode_64bit_ms": round(ta, 4),
                          "B_bytes_only_ms": round(tb, 4), "B_bytes": 2 * nb, "K7_fused_ms": round(t7, 4)}))
