"""Diagnostic: per-phase shader-clock breakdown of the wave kernels (needs csrc/libfrad_hip_stamps.so, built with
-DFRAD_WAVE_STAMPS).  usage: python tools/stamps.py [enc|dec]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frad_python_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ.get("STAMPS_LIB", "libfrad_hip_stamps.so"))
import torch
from frad_python_amd import core
what = sys.argv[1] if len(sys.argv) > 1 else "enc"
dev = torch.device("cuda:0")
F, N, C, bits = 14062, 2048, 2, 32
g = torch.Generator(device=dev).manual_seed(1)
pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
enc = core.analogue_batch(0, pcm, "s16le", F, N, C, bits, check_overflow=False)
out = torch.empty((F, N, C), dtype=torch.float64, device=dev)
fn = (lambda: core.analogue_batch(0, pcm, "s16le", F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax)) if what == "enc" \
    else (lambda: core.digital_batch(0, enc.payload, F, N, C, bits, out=out))
dll = core._lib.load().dll
buf = (ctypes.c_ulonglong * 16)()
import time
for _ in range(100): fn()
torch.cuda.synchronize()
dll.frad_debug_wave_stamps(buf, 1)
reps = 50
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): fn()
b.record(); torch.cuda.synchronize()
print('us per launch', a.elapsed_time(b) / reps * 1e3)
dll.frad_debug_wave_stamps(buf, 1)
units = buf[8]
tot = sum(buf[i] for i in range(8))
print(what, "units", units, "cycles/unit", round(tot / units))
for i in range(8):
    if buf[i]: print(f"  phase {i}: {buf[i] / units:9.0f} cycles/unit  {100 * buf[i] / tot:5.1f} %")
