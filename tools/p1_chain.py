#!/usr/bin/env python3
"""The profile-1 device chain, 20 times per clip length, for tools/profile_p1.sh (rocprofv3 kernel stats)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core
dev = torch.device("cuda:0")
N, C, hop = 2048, 2, 1920
g = torch.Generator(device=dev).manual_seed(1)
loss = 1.25 ** 20 / 19 + 0.5
for secs in (60, 600):
    n = secs * 48000
    F = (n - N) // hop + 1
    pcm = (torch.randn((n, C), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
    for _ in range(20):
        q, tq = core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)
        flat, offs = core.p1_golomb_encode_batch(q, tq)
        q2, tq2, st = core.p1_golomb_decode_batch(flat, offs, N, C)
        dec = core.p1_digital_batch(q2, tq2, N, C, 16, 48000)
        out, tail = core.p1_overlap_add(dec, 16)
    torch.cuda.synchronize()
    assert torch.equal(q, q2) and torch.equal(tq, tq2)
print("ok")
