"""Kernel-only timing of the profile-4 pack / unpack (K1 / K2) at BASELINE config 2's size (14 062 stereo frames of 2048
samples), every storage depth, s16le and f32le PCM.  FRAD_TUNE_NO_P4_WAVE=1 selects the block-per-frame mapping of K1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frad_python_amd._lib as _L
if os.environ.get('KB_LIB'): _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), os.environ['KB_LIB'])      # diagnostic builds
from frad_python_amd import core
dev = torch.device("cuda:0")
F, N, C = 14062, 2048, 2
S = F * N * C
g = torch.Generator(device=dev).manual_seed(1)
def timeit(fn, n=200):
    for _ in range(50): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for fmt in ("s16le", "f32le"):
    if fmt == "s16le": pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
    else: pcm = (torch.rand((F * N, C), generator=g, device=dev) * 1.8 - 0.9)
    isz = pcm.element_size()
    for bits in (12, 16, 24, 32, 48, 64):
        enc = core.analogue_batch(4, pcm, fmt, F, N, C, bits, check_overflow=False)
        o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
        te = timeit(lambda: core.analogue_batch(4, pcm, fmt, F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax))
        td = timeit(lambda: core.digital_batch(4, enc.payload, F, N, C, bits, out=o))
        be, bd = S * (isz + bits / 8), S * (bits / 8 + 8)
        print(json.dumps({"pcm": fmt, "bits": bits, "pack_ms": round(te, 4), "pack_frac": round(be / te / 1e6 / 8000, 3),
                          "unpack_ms": round(td, 4), "unpack_frac": round(bd / td / 1e6 / 8000, 3)}), flush=True)
