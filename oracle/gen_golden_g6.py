#!/usr/bin/env python3
"""Golden vectors G6 (round 2) -- runs the REFERENCE itself, like gen_golden.py (build container only).

  python oracle/gen_golden_g6.py   ->  tests/golden/g6_p1_more.npz

What it adds to G1-G5:
  f_*   profile 1 on float PCM (f32le / f32be / f16le): the reference does not widen floats (pcmformat.py:35), so its
        DCT and band statistics run in float32 (profile1.py:21, p1tools.py:18-33) -- q / tq from profile1.analogue
  w_*   profile 1 at compact sizes wider than a CU's LDS (10240 mono, 5120 stereo, 2560 x 5): q / tq / payload bytes /
        decoded PCM from profile1.analogue + profile1.digital
  gol_* exp_golomb_rice_encode on long Laplacian integer vectors (p1tools.py:46-60), incl. all-zero and one-value input
  ff_*  from_f64 for every integer PCM format on in-range samples (pcmformat.py:49-62)
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (loader of the reference, see its docstring)
from frad_python_amd import synth  # noqa: E402


def split(p1tools, frad):
    raw = zlib.decompress(frad, wbits=-15)
    tl = int.from_bytes(raw[:4], "big")
    return p1tools.exp_golomb_rice_decode(raw[4:4 + tl]).astype(np.int32), p1tools.exp_golomb_rice_decode(raw[4 + tl:]).astype(np.int32), raw


def main():
    fourier, pcmformat, backend, asfh, encoder, decoder = gg.load_reference()
    p1 = fourier.profile1
    p1tools = __import__("libfrad.fourier.tools.p1tools", fromlist=["x"])
    g6 = {}
    for fmt in ("f32le", "f32be", "f16le"):
        dt = pcmformat.ff_format_to_numpy_type(fmt)
        for (N, C, sr) in ((2048, 2, 48000), (640, 1, 32000)):
            raw = synth.to_pcm(synth.harmonic_mix(N, C, sr, seed=N + C) * 0.8 +
                               np.random.default_rng(N).uniform(-0.05, 0.05, (N, C)), fmt)
            g6[f"f_{fmt}_{N}_{C}_in"] = raw
            pcm = pcmformat.to_f64(np.frombuffer(raw.tobytes(), dt).reshape(-1, C), dt)
            assert pcm.dtype.kind == "f" and pcm.dtype.itemsize < 8
            for lv, ll in (("a", 0.553), ("b", 5.0)):
                frad, idx, ch, srr = p1.analogue(pcm, 16, sr, ll)
                tq, q, _ = split(p1tools, frad)
                g6[f"f_{fmt}_{N}_{C}_{lv}_q"] = q
                g6[f"f_{fmt}_{N}_{C}_{lv}_tq"] = tq
    dt = pcmformat.ff_format_to_numpy_type("s16le")
    for (N, C, sr) in ((10240, 1, 48000), (5120, 2, 44100), (2560, 5, 96000)):
        raw = synth.to_pcm(synth.harmonic_mix(N, C, sr, seed=N + C), "s16le")
        g6[f"w_{N}_{C}_in"] = raw
        frad, idx, ch, srr = p1.analogue(pcmformat.to_f64(raw, dt), 16, sr, 1.0)
        tq, q, inflated = split(p1tools, frad)
        g6[f"w_{N}_{C}_q"] = q
        g6[f"w_{N}_{C}_tq"] = tq
        g6[f"w_{N}_{C}_gol"] = gg.u8(inflated)
        g6[f"w_{N}_{C}_dec"] = p1.digital(frad, idx, ch, srr, N)
    rng = np.random.default_rng(606)
    vecs = {"lap4k": np.rint(rng.laplace(0, 6.0, 4096)).astype(np.int64),
            "lap_wide": np.rint(rng.laplace(0, 900.0, 3000)).astype(np.int64),
            "sparse": (rng.integers(0, 40, 5000) == 0) * rng.integers(-3, 4, 5000),
            "zeros": np.zeros(777, np.int64), "one": np.array([-5]), "pow2": np.array([4, -4, 8, -8, 1024, -1024, 0, 1]),
            "big": np.array([2 ** 31 - 1, -(2 ** 31) + 1, 0, 12345678, -1])}
    for name, v in vecs.items():
        v = np.asarray(v, dtype=int)
        g6[f"gol_{name}_data"] = v.astype(np.int64)
        g6[f"gol_{name}_bytes"] = gg.u8(p1tools.exp_golomb_rice_encode(v))
    ff = np.concatenate([rng.uniform(-1, 1, 56), [0.0, -1.0, 0.5, -0.5, 0.999969482421875, -0.999969482421875, 1e-9, -1e-9]])
    g6["ff_in"] = ff
    for fmt in ("u8", "u16le", "u16be", "u32le", "u32be", "s8", "s16le", "s16be", "s32le", "s32be", "s64le", "s64be", "u64le"):
        dtp = pcmformat.ff_format_to_numpy_type(fmt)
        with np.errstate(all="ignore"):
            out = pcmformat.from_f64(ff, dtp)
        g6[f"ff_{fmt}"] = gg.u8(np.ascontiguousarray(out).tobytes())
        g6[f"ff_{fmt}_dtype"] = np.array(out.dtype.str)
    np.savez_compressed(os.path.join(gg.OUT, "g6_p1_more.npz"), **g6)
    print("g6:", len(g6), "arrays,", os.path.getsize(os.path.join(gg.OUT, "g6_p1_more.npz")), "bytes")


if __name__ == "__main__":
    main()
