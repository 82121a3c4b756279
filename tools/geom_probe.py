import torch, sys
sys.path.insert(0, '.')
from frad_python_amd import core
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
g = torch.Generator(device=dev).manual_seed(1)
S = 2812 * 4096 * 8
for (N, C, fmt, bits) in [(4096, 8, "f32le", 32), (2048, 8, "f32le", 32), (4096, 4, "f32le", 32), (4096, 2, "f32le", 32), (2048, 2, "f32le", 32), (8192, 8, "f32le", 32), (4096, 8, "s16le", 16), (4096, 8, "s32le", 64),
                          (4096, 2, "s16le", 32), (1024, 2, "s16le", 32), (8192, 2, "s16le", 32), (512, 2, "s16le", 32), (2048, 1, "s16le", 32), (2048, 6, "s16le", 24)]:
    F = S // (N * C)
    if fmt == "f32le": pcm = (torch.rand((F * N, C), generator=g, device=dev) * 1.8 - 0.9).to(torch.float32)
    elif fmt == "s16le": pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
    else: pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).to(torch.int32)
    enc = core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False)
    o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
    te = timeit(lambda: core.analogue_batch(0, pcm, fmt, F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax))
    td = timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, bits, out=o))
    isz = pcm.element_size()
    print(f"N={N} C={C} {fmt} b{bits}: enc {te:.3f} ms ({S*(isz+bits/8)/te/1e6:.0f} GB/s)  dec {td:.3f} ms ({S*(bits/8+8)/td/1e6:.0f} GB/s)", flush=True)
