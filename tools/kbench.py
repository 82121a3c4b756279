"""Kernel-only timing of the headline launches (cfg 2: 14 062 stereo s16 frames, N = 2048, 32-bit BE), steady clocks.
usage: python tools/kbench.py [enc|dec|both] [reps]   -- prints mean us per launch over `reps` back-to-back launches"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frad_python_amd._lib as _L
if os.environ.get('KB_LIB'): _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), os.environ['KB_LIB'])
from frad_python_amd import core
what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fmt = os.environ.get("KB_FMT", "s16le"); C = int(os.environ.get("KB_C", "2")); bits = int(os.environ.get("KB_BITS", "32"))
dev = torch.device("cuda:0")
F, N = 14062, 2048
g = torch.Generator(device=dev).manual_seed(1)
if fmt.startswith("s16"): pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
elif fmt.startswith("s32"): pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000 * 65536).to(torch.int32)
elif fmt.startswith("f64"): pcm = (torch.rand((F * N, C), generator=g, device=dev, dtype=torch.float64) * 1.8 - 0.9)
elif fmt.startswith("u8"): pcm = (torch.rand((F * N, C), generator=g, device=dev) * 255).to(torch.uint8)
enc = core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False)
out = torch.empty((F, N, C), dtype=torch.float64, device=dev)
fe = lambda: core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax)
fd = lambda: core.digital_batch(0, enc.payload, F, N, C, bits, out=out)
def timeit(fn, n):
    for _ in range(150): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
S = F * N * C
isz = pcm.element_size()
if what in ("enc", "both"):
    t = timeit(fe, reps); print(f"enc {fmt} C={C} b{bits}: {t:.1f} us  {S*(isz+bits/8)/t/1e6:.2f} TB/s  frac {S*(isz+bits/8)/t/1e6/8:.3f}", flush=True)
if what in ("dec", "both"):
    t = timeit(fd, reps); print(f"dec {fmt} C={C} b{bits}: {t:.1f} us  {S*(bits/8+8)/t/1e6:.2f} TB/s  frac {S*(bits/8+8)/t/1e6/8:.3f}", flush=True)
