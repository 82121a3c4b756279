import torch, sys, os
sys.path.insert(0, '/root/repo')
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
for (N, C) in [(512, 2), (256, 2), (128, 2), (512, 1)]:
    F = S // (N * C)
    pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
    enc = core.analogue_batch(0, pcm, "s16le", F, N, C, 32, check_overflow=False)
    o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
    te = timeit(lambda: core.analogue_batch(0, pcm, "s16le", F, N, C, 32, check_overflow=False, out=enc.payload, absmax=enc.absmax))
    td = timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=o))
    print(os.environ.get("FRAD_TUNE_FPB"), f"N={N} C={C}: enc {te:.3f} ms ({S*6/te/1e6:.0f} GB/s)  dec {td:.3f} ms ({S*12/td/1e6:.0f} GB/s)", flush=True)
