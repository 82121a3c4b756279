#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (not the headline bench): kernel-only HIP-event timings of the other
BASELINE configurations -- profile 4 pack/unpack at the cfg-2 size, cfg 4 (192 kHz 7.1 f32, N=4096), cfg 5
(profile 1 quantiser, N=2048 hop 1920) -- as algorithmic GB/s and fraction of the 8 TB/s HBM peak."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def line(name, nbytes, ms, samples):
    gbs = nbytes / ms / 1e6
    return {"case": name, "ms": round(ms, 4), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / 8000, 4),
            "Gsamples/s": round(samples / ms / 1e6, 2)}


out = []
g = torch.Generator(device=dev).manual_seed(1)
# profile 4 (pure cast + pack), cfg-2 sized, s16 -> 32-bit and 24-bit
F, N, C = 14062, 2048, 2
pcm = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
S = F * N * C
for bits in (32, 24, 16):
    enc = core.analogue_batch(4, pcm, "s16le", F, N, C, bits, check_overflow=False)
    o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
    out.append(line(f"p4 encode s16->b{bits}", S * (2 + bits / 8), timeit(lambda: core.analogue_batch(4, pcm, "s16le", F, N, C, bits, check_overflow=False, out=enc.payload, absmax=enc.absmax)), S))
    out.append(line(f"p4 decode b{bits}->f64", S * (bits / 8 + 8), timeit(lambda: core.digital_batch(4, enc.payload, F, N, C, bits, out=o)), S))
# profile 0 decode with the from_f64 conversion fused into the wave kernel (frad_p0_digital_pcm): B_out = itemsize
enc0 = core.analogue_batch(0, pcm, "s16le", F, N, C, 32, check_overflow=False)
for ofmt, osz in (("s16le", 2), ("f32le", 4)):
    ob = torch.empty(F * N * C * osz, dtype=torch.uint8, device=dev)
    out.append(line(f"p0 decode b32->{ofmt} (conversion fused, cfg-2 size)", S * (4 + osz),
                    timeit(lambda: core.digital_batch(0, enc0.payload, F, N, C, 32, out=ob, out_format=ofmt)), S))
o0 = torch.empty((F, N, C), dtype=torch.float64, device=dev)
out.append(line("p0 decode b32->f64 (cfg-2 size, for comparison)", S * 12, timeit(lambda: core.digital_batch(0, enc0.payload, F, N, C, 32, out=o0)), S))
# the same samples as 28 124 mono frames (two frames per wave)
pcm_m = pcm.reshape(-1, 1)
enc_m = core.analogue_batch(0, pcm_m, "s16le", 2 * F, N, 1, 32, check_overflow=False)
ob = torch.empty(S * 2, dtype=torch.uint8, device=dev)
out.append(line("p0 decode b32->s16le, mono (conversion fused)", S * 6, timeit(lambda: core.digital_batch(0, enc_m.payload, 2 * F, N, 1, 32, out=ob, out_format="s16le")), S))
om = torch.empty((2 * F, N, 1), dtype=torch.float64, device=dev)
out.append(line("p0 decode b32->f64, mono (for comparison)", S * 12, timeit(lambda: core.digital_batch(0, enc_m.payload, 2 * F, N, 1, 32, out=om)), S))
del pcm_m, enc_m, om
# cfg 4: 60 s of 192 kHz 8-channel f32, N = 4096, 32 bit
F, N, C = 2812, 4096, 8
pcm4 = (torch.rand((F * N, C), generator=g, device=dev) * 1.8 - 0.9).to(torch.float32)
S = F * N * C
enc = core.analogue_batch(0, pcm4, "f32le", F, N, C, 32, check_overflow=False)
o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
out.append(line("cfg4 p0 encode f32 8ch N=4096 (f32 compute, half-frame blocks)", S * 8, timeit(lambda: core.analogue_batch(0, pcm4, "f32le", F, N, C, 32, check_overflow=False, out=enc.payload, absmax=enc.absmax)), S))
out.append(line("cfg4 p0 decode (f64, channel-group kernel)", S * 12, timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=o)), S))
for fmt, bo in (("f32le", 4), ("s16le", 2)):               # the conversion in the kernel's own store (no float64 scratch, round 3)
    on = core._pcm_out_tensor(fmt, (F, N, C), dev)
    out.append(line(f"cfg4 p0 decode -> {fmt} (converting store)", S * (4 + bo), timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=on, out_format=fmt)), S))
del pcm4, enc, o, on
# N = 1024 stereo (the unit kernel), 28 126 frames: float64 out and s16 out (converting store)
F, N, C = 28126, 1024, 2
S = F * N * C
pcm1 = (torch.randn((F * N, C), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
enc = core.analogue_batch(0, pcm1, "s16le", F, N, C, 32, check_overflow=False)
o = torch.empty((F, N, C), dtype=torch.float64, device=dev)
out.append(line("N=1024 stereo p0 decode (f64, unit kernel)", S * 12, timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=o)), S))
on = core._pcm_out_tensor("s16le", (F, N, C), dev)
out.append(line("N=1024 stereo p0 decode -> s16le (converting store)", S * 6, timeit(lambda: core.digital_batch(0, enc.payload, F, N, C, 32, out=on, out_format="s16le")), S))
del pcm1, enc, o, on
# cfg 3: one GPU's share (512) of 4096 x 1 s stereo clips: 23 full frames + an 896-sample tail frame per clip
clips, n3 = 512, 48000
full3, tail3 = n3 // 2048, n3 % 2048
pcm3 = (torch.randn((clips, n3, 2), generator=g, device=dev) * 8000).clamp(-32768, 32767).to(torch.int16)
body3, tails3 = pcm3[:, :full3 * 2048].contiguous(), pcm3[:, full3 * 2048:].contiguous()
F3 = clips * full3
enc3 = core.analogue_batch(0, body3, "s16le", F3, 2048, 2, 32, check_overflow=False)
o3 = torch.empty((F3, 2048, 2), dtype=torch.float64, device=dev)
S3 = F3 * 2048 * 2
out.append(line("cfg3 full frames encode (11776 x N=2048)", S3 * 6, timeit(lambda: core.analogue_batch(0, body3, "s16le", F3, 2048, 2, 32, check_overflow=False, out=enc3.payload, absmax=enc3.absmax)), S3))
out.append(line("cfg3 full frames decode", S3 * 12, timeit(lambda: core.digital_batch(0, enc3.payload, F3, 2048, 2, 32, out=o3)), S3))
et3 = core.analogue_batch(0, tails3, "s16le", clips, tail3, 2, 32, check_overflow=False)
ot3 = torch.empty((clips, tail3, 2), dtype=torch.float64, device=dev)
St = clips * tail3 * 2
out.append(line("cfg3 tail frames encode (512 x N=896 = 7 x 128, mixed-radix kernels)", St * 6, timeit(lambda: core.analogue_batch(0, tails3, "s16le", clips, tail3, 2, 32, check_overflow=False, out=et3.payload, absmax=et3.absmax)), St))
out.append(line("cfg3 tail frames decode (512 x N=896 = 7 x 128, mixed-radix kernels)", St * 12, timeit(lambda: core.digital_batch(0, et3.payload, clips, tail3, 2, 32, out=ot3)), St))
# cfg 5: 60 s stereo s16, profile 1, N = 2048, hop 1920, loss level 20
N, C, hop = 2048, 2, 1920
n = 60 * 48000
F = (n - N) // hop + 1
pcm5 = (torch.randn((n, C), generator=g, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
loss = 1.25 ** 20 / 19 + 0.5
q, tq = core.p1_analogue_batch(pcm5, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)
out.append(line("cfg5 p1 quantise (K7)", F * (N * C * 2 + N * C * 4 + 27 * C * 4), timeit(lambda: core.p1_analogue_batch(pcm5, "s16le", F, N, C, 16, 48000, loss, frame_stride=hop)), F * hop * C))
out.append(line("cfg5 p1 dequantise+IDCT (K8)", F * (N * C * 4 + 27 * C * 4 + N * C * 8), timeit(lambda: core.p1_digital_batch(q, tq, N, C, 16, 48000)), F * hop * C))
dec = core.p1_digital_batch(q, tq, N, C, 16, 48000)
out.append(line("cfg5 overlap-add", F * (N * C * 8 + hop * C * 8), timeit(lambda: core.p1_overlap_add(dec, 16)), F * hop * C))
# Exp-Golomb-Rice stage on the device (row 8f #2): bytes = what the coder reads + writes (int32 in, body bytes out)
flat, offs = core.p1_golomb_encode_batch(q, tq)
body_total = int(offs[-1].item())
out.append(line("cfg5 Exp-Golomb encode (+ compaction)", F * (N * C * 4 + 27 * C * 4) + 2 * body_total,
                timeit(lambda: core.p1_golomb_encode_batch(q, tq)), F * hop * C))
out.append(line("cfg5 Exp-Golomb decode", F * (N * C * 4 + 27 * C * 4) + body_total,
                timeit(lambda: core.p1_golomb_decode_batch(flat, offs, N, C)), F * hop * C))
out[-1]["body_bytes_per_frame"] = round(body_total / F, 1)
# device-copy microbench (what a plain copy reaches of the nominal 8 TB/s on this box): read + write bytes / time
for mib in (256, 1024):
    a = torch.empty(mib << 20, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    ms = timeit(lambda: b.copy_(a))
    out.append({"case": f"device copy {mib} MiB (read + write)", "ms": round(ms, 4), "GB/s": round(2 * (mib << 20) / ms / 1e6, 1),
                "hbm_frac": round(2 * (mib << 20) / ms / 1e6 / 8000, 4)})
    del a, b
# end to end through the streaming API (SURVEY 8d "separate line"): host bytes in -> FrAD stream bytes out and
# back, i.e. H2D + kernels + D2H + the Python ASFH framer / CRC, one process() call per 60 s of audio
import time  # noqa: E402
from frad_python_amd import Decoder, Encoder  # noqa: E402
secs = 60
host = (np.random.default_rng(3).normal(0, 3000, (secs * 48000, 2)).clip(-32768, 32767)).astype("<i2").tobytes()
best_e, best_d = 1e9, 1e9
for _ in range(4):                                            # best of 4: the first pass pays allocator and table set-up
    t0 = time.perf_counter()
    enc_s = Encoder(0, 48000, 2, 32, 2048, "s16le")
    r = enc_s.process(host); tail_b = enc_s.flush().buf
    t1 = time.perf_counter()
    stream = r.buf + tail_b
    t1b = time.perf_counter()
    dec_s = Decoder()
    d = dec_s.process(stream); tail = dec_s.flush()
    t2 = time.perf_counter()
    best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1b)
t0, t1, t2 = 0.0, best_e, best_e + best_d
S2 = secs * 48000 * 2
out.append({"case": "e2e stream encode, host bytes -> FrAD bytes (60 s stereo s16, profile 0, 32 bit)", "ms": round((t1 - t0) * 1e3, 2),
            "Gsamples/s": round(S2 / (t1 - t0) / 1e9, 3), "stream_bytes": len(stream)})
out.append({"case": "e2e stream decode, FrAD bytes -> host float64", "ms": round((t2 - t1) * 1e3, 2),
            "Gsamples/s": round(S2 / (t2 - t1) / 1e9, 3), "frames": int(d.frames)})
# decode with the output conversion on the device (Decoder(out_format="s16le")): 2 bytes per sample over PCIe instead of 8
best_n = 1e9
for _ in range(4):
    t1b = time.perf_counter()
    dec_s = Decoder(out_format="s16le")
    dn = dec_s.process(stream); dec_s.flush()
    best_n = min(best_n, time.perf_counter() - t1b)
out.append({"case": "e2e stream decode, FrAD bytes -> host s16 (from_f64 on the device)", "ms": round(best_n * 1e3, 2),
            "Gsamples/s": round(S2 / best_n / 1e9, 3), "frames": int(dn.frames), "dtype": str(dn.pcm.dtype)})
# the same for profile 1 (cfg 5): quantiser + Golomb coder on the device, deflate + ASFH on the host
host5 = pcm5.cpu().numpy().tobytes()
best_e, best_d = 1e9, 1e9
for _ in range(3):
    t0 = time.perf_counter()
    enc_s = Encoder(1, 48000, 2, 16, 2048, "s16le"); enc_s.set_overlap_ratio(16); enc_s.set_loss_level(loss)
    r = enc_s.process(host5); tail_b = enc_s.flush().buf
    t1 = time.perf_counter()
    stream5 = r.buf + tail_b
    t1b = time.perf_counter()
    dec_s = Decoder()
    d5 = dec_s.process(stream5); tail = dec_s.flush()
    t2 = time.perf_counter()
    best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1b)
out.append({"case": "e2e stream encode, profile 1 (60 s stereo s16, N=2048, overlap 16, loss level 20)", "ms": round(best_e * 1e3, 2),
            "Gsamples/s": round(S2 / best_e / 1e9, 4), "stream_bytes": len(stream5)})
out.append({"case": "e2e stream decode, profile 1", "ms": round(best_d * 1e3, 2), "Gsamples/s": round(S2 / best_d / 1e9, 4), "frames": int(d5.frames)})
for o_ in out:
    print(json.dumps(o_))
