import torch, sys
sys.path.insert(0, '.')
from frad_python_amd import core
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
g = torch.Generator(device=dev).manual_seed(1)
clips, n, C, N = 512, 48000, 2, 2048
full, tail = n // N, n % N
pcm = (torch.randn((clips, n, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
body = pcm[:, :full * N].contiguous(); tails = pcm[:, full * N:].contiguous()
F = clips * full
enc = core.analogue_batch(0, body, "s16le", F, N, C, 32, check_overflow=False)
o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
print("main enc", timeit(lambda: core.analogue_batch(0, body, "s16le", F, N, C, 32, check_overflow=False, out=enc.payload, absmax=enc.absmax)))
print("main dec", timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=o)))
for T in (tail, 1024, 512, 896 // 7 * 8):
    tt = pcm[:, :T].contiguous()
    et = core.analogue_batch(0, tt, "s16le", clips, T, C, 32, check_overflow=False)
    ot = torch.empty((clips, T, C), dtype=torch.float64, device=dev)
    print("tail N=%d enc" % T, timeit(lambda: core.analogue_batch(0, tt, "s16le", clips, T, C, 32, check_overflow=False, out=et.payload, absmax=et.absmax)),
          "dec", timeit(lambda: core.digital_batch(0, et.payload, clips, T, C, 32, out=ot)))
