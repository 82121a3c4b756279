#!/usr/bin/env python3
"""A few launches of every kernel OFF the headline path, for the PMC passes of tools/profile_round.sh (rocprofv3 --pmc):
K7 / K8 on the 10-minute cfg-5 clip, the Exp-Golomb coder / decoder on the same frames, cfg 4 encode / decode (2 812 frames of
192 kHz 7.1 float32, N = 4096), cfg 3's 896-sample tail frames of 512 clips in place (mixed-radix kernels), the overlap-add,
profile-4 pack / unpack at 12 / 16 / 24 bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
REPS = int(os.environ.get("PMC_REPS", "4"))
# cfg 5: K7, Golomb, K8, overlap-add
N, C, hop = 2048, 2, 1920
n = 600 * 48000
F = (n - N) // hop + 1
pcm = (torch.randn((n, C), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
loss = 1.25 ** 20 / 19 + 0.5
for _ in range(REPS):
    q, tq = core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)
    flat, offs = core.p1_golomb_encode_batch(q, tq)
    q2, tq2, st = core.p1_golomb_decode_batch(flat, offs, N, C)
    dec = core.p1_digital_batch(q2, tq2, N, C, 16, 48000)
    out, tail = core.p1_overlap_add(dec, 16)
torch.cuda.synchronize()
del pcm, q, tq, q2, tq2, dec, out, flat
# cfg 4
N4, C4, F4 = 4096, 8, 2812
x = (torch.rand((F4 * N4, C4), generator=g, device=dev) * 1.8 - 0.9).to(torch.float32)
for _ in range(REPS):
    enc = core.analogue_batch(0, x, "f32le", F4, N4, C4, 32, check_overflow=False)
    d4 = core.digital_batch(0, enc.payload, F4, N4, C4, 32)
torch.cuda.synchronize()
del x, enc, d4
# cfg 3 tails: 512 clips of 48 000 sample-frames, the 896-sample last frame of each, in place
clips = (torch.randn((512, 48000, 2), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
whole = torch.empty((512, 48000, 2), dtype=torch.float64, device=dev)
for _ in range(REPS):
    et = core.analogue_clips(clips, "s16le", 896, 32, first=23 * 2048)
    core.digital_clips(et.payload, whole, 896, 32, first=23 * 2048)
torch.cuda.synchronize()
del clips, whole, et
# profile 4 (K1 / K2) at cfg 2's size: 16-bit (wave per frame), 24-bit (48-byte units through LDS, 6-byte pairs), 12-bit (3-byte pairs)
F2, N2, C2 = 14062, 2048, 2
p2 = (torch.randn((F2 * N2, C2), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
for _ in range(REPS):
    for bits in (16, 24, 12):
        e4 = core.analogue_batch(4, p2, "s16le", F2, N2, C2, bits, check_overflow=False)
        d2 = core.digital_batch(4, e4.payload, F2, N2, C2, bits)
torch.cuda.synchronize()
print("ok")
