"""Diagnostic: per-phase shader-clock breakdown of the wave kernels (needs csrc/libfrad_hip_stamps.so:
tools/build_variant.sh stamps -DFRAD_WAVE_STAMPS frad_p0_wave frad_p1_wave).  usage: python tools/stamps.py [enc|dec|k7]
phases: 0 wait for the PCM DMA, 1 Makhoul pairs + conversion, 2 pass 1, 3 twiddle + exchange, 4 pass 2, 5 pair step + pack + stores (p0),
6 loop end; K7 tail: 7 pair step + plane round 0, 8 band sums, 9 thresholds, 6 quantiser + stores"""
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
if what == "k7":
    fn = lambda: core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, 1.25 ** 20 / 19 + 0.5)
dll = core._lib.load().dll
stamps = dll.frad_debug_wave_stamps_p1 if what == "k7" else dll.frad_debug_wave_stamps
buf = (ctypes.c_ulonglong * 16)()
import time
for _ in range(100): fn()
torch.cuda.synchronize()
stamps(buf, 1)
reps = 50
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): fn()
b.record(); torch.cuda.synchronize()
print('us per launch', a.elapsed_time(b) / reps * 1e3)
stamps(buf, 1)
units = buf[15]
tot = sum(buf[i] for i in range(12))
print(what, "units", units, "cycles/unit", round(tot / units))
for i in range(12):
    if buf[i]: print(f"  phase {i}: {buf[i] / units:9.0f} cycles/unit  {100 * buf[i] / tot:5.1f} %")
