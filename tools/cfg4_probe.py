#!/usr/bin/env python3
"""Probe of the two-pass kernels for frames whose float64 channels need two passes through a CU's LDS (k_p0_fwd_grp2 /
k_p0_inv_grp2): cfg 4's geometry (192 kHz 7.1, N = 4096) and its neighbours; one JSON line per geometry and direction.
A/B knobs: FRAD_TUNE_GRP2_WHOLE=1 (one 4 + 4 block per frame instead of two 2 + 2 blocks), FRAD_TUNE_GRP2_PIPE=1 (the
persistent software-pipelined decode), FRAD_PROBE_LIB (a diagnostic build of libfrad_hip.so), CFG4_GEOMS (first n geometries)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core, _lib  # noqa: E402

if os.environ.get("FRAD_PROBE_LIB"):                     # diagnostic builds of libfrad_hip.so (kernel ablations)
    _lib.LIB_PATH = os.environ["FRAD_PROBE_LIB"]
dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


g = torch.Generator(device=dev).manual_seed(1)
GEOMS = ((2812, 4096, 8, 32, "f32le"), (2812, 4096, 8, 16, "s16le"), (2812, 4096, 8, 64, "s32le"),
         (2812, 2048, 16, 32, "s16le"), (2812, 8192, 4, 32, "s16le"))
knobs = {k: os.environ[k] for k in ("FRAD_TUNE_GRP2_WHOLE", "FRAD_TUNE_GRP2_PIPE", "FRAD_PROBE_LIB") if k in os.environ}
for (F, N, C, bits, fmt) in GEOMS[:int(os.environ.get("CFG4_GEOMS", "5"))]:
    S = F * N * C
    if fmt == "f32le":
        pcm = (torch.rand((F * N, C), generator=g, device=dev) * 1.8 - 0.9).to(torch.float32)
    elif fmt == "s16le":
        pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
    else:
        pcm = (torch.randn((F * N, C), generator=g, device=dev) * 5e8).clamp(-2**31, 2**31 - 1).to(torch.int32)
    enc = core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False)
    o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
    ms = timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, bits, out=o))
    nb = S * (bits / 8 + 8)
    print(json.dumps({"case": f"p0 decode N={N} C={C} b{bits}", "ms": round(ms, 4), "GB/s": round(nb / ms / 1e6, 1),
                      "hbm_frac": round(nb / ms / 8e9, 4), "knobs": knobs}), flush=True)
    ems = timeit(lambda: core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax))
    enb = S * (bits / 8 + pcm.element_size())
    print(json.dumps({"case": f"p0 encode {fmt} N={N} C={C} b{bits}", "ms": round(ems, 4), "GB/s": round(enb / ems / 1e6, 1),
                      "hbm_frac": round(enb / ems / 8e9, 4), "knobs": knobs}), flush=True)
