#!/usr/bin/env python3
"""Golden-vector generator -- runs the REFERENCE itself (build container only).

Imports H4n-uL/FrAD_Python from /root/reference (read-only, never copied, never shipped to
the GPU box), feeds it seeded inputs and stores inputs + the reference's outputs as small
data fixtures under tests/golden/.  Re-run with:  python oracle/gen_golden.py

How the reference is loaded:
 * ``libfrad.fourier.*`` and ``libfrad.tools.asfh`` import cleanly once ``libfrad`` is
   registered as a bare package (its ``__init__`` is skipped because it pulls in the
   Reed-Solomon wrapper, whose third-party module ``reedsolo`` is not installed here).
 * For the whole-stream fixture (G3) the reference ``Encoder``/``Decoder`` classes are needed;
   they import ``tools/ecc.py`` -> ``reedsolo`` at module level.  ECC is off by default and is
   never invoked, so the generator registers an inert placeholder module for that one import.
 * Python 3.10's ``zlib.compress`` has no ``wbits=`` keyword (the reference's profile 1
   needs >= 3.11): the generator rebinds ``profile1.zlib`` to a wrapper that forwards
   ``wbits`` to ``zlib.compressobj``, which is what 3.11 does internally.
Nothing in the reference tree is modified.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/src"
sys.path.insert(0, ROOT)

from frad_python_amd import synth  # noqa: E402


def load_reference():
    sys.path.insert(0, REF)
    pkg = types.ModuleType("libfrad")
    pkg.__path__ = [os.path.join(REF, "libfrad")]
    sys.modules["libfrad"] = pkg
    import libfrad.fourier as fourier                          # noqa
    import libfrad.backend.pcmformat as pcmformat              # noqa
    import libfrad.backend as backend                          # noqa
    import libfrad.tools.asfh as asfh                          # noqa

    class _Z:                                                  # py3.10 zlib.compress(wbits=)
        error = zlib.error
        decompress = staticmethod(zlib.decompress)

        @staticmethod
        def compress(data, level=-1, wbits=15):
            co = zlib.compressobj(level, zlib.DEFLATED, wbits)
            return co.compress(data) + co.flush()
    fourier.profile1.zlib = _Z

    placeholder = types.ModuleType("reedsolo")                 # ECC is never called (off)

    class _Unavailable:
        def __init__(self, *a, **k):
            raise RuntimeError("reedsolo is not installed; ECC is outside the golden vectors")
    placeholder.RSCodec = _Unavailable
    placeholder.ReedSolomonError = type("ReedSolomonError", (Exception,), {})
    sys.modules["reedsolo"] = placeholder
    import libfrad.encoder as encoder                          # noqa
    import libfrad.decoder as decoder                          # noqa
    return fourier, pcmformat, backend, asfh, encoder, decoder


def u8(b: bytes) -> np.ndarray:
    return np.frombuffer(b, np.uint8).copy()


def main():
    fourier, pcmformat, backend, asfh, encoder, decoder = load_reference()
    p0, p1, p4 = fourier.profile0, fourier.profile1, fourier.profile4
    p1tools = fourier.tools.p1tools if hasattr(fourier, "tools") else __import__(
        "libfrad.fourier.tools.p1tools", fromlist=["x"])
    os.makedirs(OUT, exist_ok=True)
    manifest = {"numpy": np.__version__, "scipy": __import__("scipy").__version__,
                "python": sys.version.split()[0]}

    # ------------------------------------------------------------------ G1: pack / unpack
    g1 = {"input": [1 / 3, -(0.5 ** 0.5), 1.0, 1e-5, 0.1], "cases": []}
    x5 = np.array(g1["input"]).reshape(-1, 1)
    for n in (5, 4, 1):
        for bits in p4.DEPTHS:
            for le in (False, True):
                frad, idx, ch, sr = p4.analogue(x5[:n], bits, 48000, le)
                dec = p4.digital(frad, idx, 1, le)
                g1["cases"].append({"n": n, "bits": bits, "le": le, "hex": frad.hex(), "idx": idx,
                                    "decoded_hex": dec.astype("<f8").tobytes().hex()})
    json.dump(g1, open(os.path.join(OUT, "g1_pack.json"), "w"), indent=0)

    # ------------------------------------------------------------------ G2: profile 0 / 4 frames
    g2 = {}
    index = []
    rng = np.random.default_rng(1234)
    small = [(4, 1), (4, 2), (7, 3), (16, 2)]
    medium = [(896, 2), (1024, 1), (2048, 2)]
    large = [(4096, 8), (4096, 2), (128, 1), (8192, 1)]

    def run_case(tag, N, C, fmt, bits, le, x_float, profile, store_decoded=True):
        dt = pcmformat.ff_format_to_numpy_type(fmt)
        raw = synth.to_pcm(x_float, fmt)
        frame = pcmformat.to_f64(raw.reshape(-1, C), dt)
        mod = p0 if profile == 0 else p4
        frad, idx, ch, sr = mod.analogue(frame, bits, 48000, le)
        dec = mod.digital(frad, idx, ch, le)
        key = f"{tag}_p{profile}_{N}x{C}_{fmt}_b{bits}_{'le' if le else 'be'}"
        g2[key + "_in"] = raw.view(np.uint8).reshape(-1) if raw.dtype.byteorder == ">" else raw
        g2[key + "_frad"] = u8(frad)
        if store_decoded:
            g2[key + "_dec"] = dec
        index.append({"key": key, "N": N, "C": C, "fmt": fmt, "bits": bits, "le": le,
                      "profile": profile, "idx": idx, "dec": store_decoded,
                      "dec_sha256": hashlib.sha256(dec.astype("<f8").tobytes()).hexdigest()})

    for N, C in small:
        xu = rng.uniform(-1, 1, (N, C))
        for fmt in ("s16le", "f32le", "f64le", "u8", "s32le", "f16le", "f64be", "s16be"):
            for bits in p0.DEPTHS:
                for le in (False, True):
                    for prof in (0, 4):
                        run_case("u", N, C, fmt, bits, le, xu, prof)
    xo = np.ones((4, 2)); xo[:, 1] = np.arange(4)          # payload-order probe
    run_case("order", 4, 2, "f64le", 64, False, xo / 4, 0)
    for N, C in medium:
        xu = rng.uniform(-1, 1, (N, C))
        xs = synth.harmonic_mix(N, C, 48000, seed=5)
        for fmt in ("s16le", "f32le", "f64le"):
            for bits in p0.DEPTHS:
                run_case("u", N, C, fmt, bits, False, xu, 0, store_decoded=bits in (12, 32, 64))
            run_case("h", N, C, fmt, 32, False, xs, 0)
            run_case("u", N, C, fmt, 24, True, xu, 0, store_decoded=False)
            run_case("u", N, C, fmt, 32, False, xu, 4, store_decoded=False)
        run_case("u", N, C, "s16le", 48, True, xu, 0, store_decoded=False)
    for N, C in large:
        xu = rng.uniform(-1, 1, (N, C)) * 0.9
        run_case("u", N, C, "f32le", 32, False, xu, 0, store_decoded=(C <= 2))
        run_case("u", N, C, "s16le", 32, False, xu, 0, store_decoded=(C <= 2))
        if C <= 2:
            run_case("u", N, C, "f64le", 64, False, xu, 0)
    np.savez_compressed(os.path.join(OUT, "g2_frames.npz"), **g2)
    json.dump(index, open(os.path.join(OUT, "g2_index.json"), "w"), indent=0)

    # ------------------------------------------------------------------ G3: whole streams
    g3 = {"cases": []}

    def stream_case(name, pcm_bytes, chunk, **kw):
        enc = encoder.Encoder(kw["profile"], kw["srate"], kw["channels"], kw["bits"],
                              kw["frame_size"], kw["pcm_format"])
        enc.set_little_endian(kw.get("little_endian", False))
        enc.set_overlap_ratio(kw.get("overlap_ratio", 0))
        enc.set_loss_level(kw.get("loss_level", 0.5))
        out = b""
        samples = 0
        for i in range(0, len(pcm_bytes), chunk):
            r = enc.process(pcm_bytes[i:i + chunk]); out += r.buf; samples += r.samples
        r = enc.flush(); out += r.buf; samples += r.samples
        dec = decoder.Decoder()
        pcm, frames = [], 0
        for i in range(0, len(out), chunk):
            d = dec.process(out[i:i + chunk]); pcm.append(d.pcm.reshape(-1, kw["channels"])); frames += d.frames
        d = dec.flush(); pcm.append(d.pcm.reshape(-1, kw["channels"]))
        pcm = np.concatenate(pcm)
        first_len = 32 + int.from_bytes(out[4:8], "big") if kw["profile"] in (0, 4) else 0
        case = dict(name=name, params=kw, nbytes=len(out), sha256=hashlib.sha256(out).hexdigest(),
                    samples=samples, frames=frames, decoded_shape=list(pcm.shape),
                    decoded_sha256=hashlib.sha256(np.ascontiguousarray(pcm).astype("<f8").tobytes()).hexdigest(),
                    first_frame_hex=out[:first_len].hex() if first_len and first_len <= 600 else out[:64].hex())
        return case, out, pcm

    sine1s = synth.sine(48000, 1, 48000, 440.0, 0.5).astype(">f8").tobytes()
    for prof in (0, 4):
        for bits in (16, 32, 64):
            c, out, pcm = stream_case(f"cfg1_p{prof}_b{bits}", sine1s, 32768, profile=prof, srate=48000,
                                      channels=1, bits=bits, frame_size=2048, pcm_format="f64be")
            g3["cases"].append(c)
    tiny = np.array([0.25, -0.5, 0.75, 0.125]).astype(">f8").tobytes()
    c, out, pcm = stream_case("tiny_p0_b16", tiny, 32768, profile=0, srate=48000, channels=1, bits=16,
                              frame_size=4, pcm_format="f64be")
    c["stream_hex"] = out.hex(); g3["cases"].append(c)
    st = synth.to_pcm(synth.harmonic_mix(3000, 2, 44100, seed=3), "s16le").tobytes()
    for prof, bits, le in ((0, 24, True), (4, 12, False), (0, 48, False)):
        c, out, pcm = stream_case(f"st_p{prof}_b{bits}", st, 1000, profile=prof, srate=44100, channels=2,
                                  bits=bits, frame_size=1024, pcm_format="s16le", little_endian=le)
        g3["cases"].append(c)
    g3_arr = {}
    sig = synth.to_pcm(synth.harmonic_mix(3 * 1920 + 700, 2, 48000, seed=1234), "s16le")
    for lv in (0, 10, 20):
        ll = 1.25 ** lv / 19.0 + 0.5
        c, out, pcm = stream_case(f"p1_lv{lv}", sig.tobytes(), 4096, profile=1, srate=48000, channels=2,
                                  bits=16, frame_size=2048, pcm_format="s16le", overlap_ratio=16, loss_level=ll)
        g3["cases"].append(c)
        g3_arr[f"p1_lv{lv}_stream"] = u8(out)
        g3_arr[f"p1_lv{lv}_decoded"] = pcm
    g3_arr["p1_input_s16le"] = sig
    json.dump(g3, open(os.path.join(OUT, "g3_streams.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(OUT, "g3_p1_streams.npz"), **g3_arr)

    # ------------------------------------------------------------------ G4: profile 1 pre-entropy
    g4 = {}
    frames = np.stack([sig[i * 1920:i * 1920 + 2048] for i in range(3)])          # hop 1920
    g4["frames_s16le"] = frames
    dt = pcmformat.ff_format_to_numpy_type("s16le")
    for lv in (0, 10, 20):
        ll = 1.25 ** lv / 19.0 + 0.5
        for i, fr in enumerate(frames):
            f64 = pcmformat.to_f64(fr, dt)
            frad, idx, ch, sr = p1.analogue(f64, 16, 48000, ll)
            raw = zlib.decompress(frad, wbits=-15)
            tl = int.from_bytes(raw[:4], "big")
            tq = p1tools.exp_golomb_rice_decode(raw[4:4 + tl])
            q = p1tools.exp_golomb_rice_decode(raw[4 + tl:])
            dec = p1.digital(frad, idx, ch, sr, 2048)
            g4[f"lv{lv}_f{i}_q"] = q.astype(np.int32)
            g4[f"lv{lv}_f{i}_tq"] = tq.astype(np.int32)
            g4[f"lv{lv}_f{i}_frad"] = u8(frad)
            g4[f"lv{lv}_f{i}_dec"] = dec
            if lv == 20:   # float intermediates straight from the reference's tool functions
                freqs = np.array([__import__("scipy.fft").fft.dct(f64[:, c], norm="forward") for c in range(2)])
                th = np.array([p1tools.mask_thres_mos(freqs[c] * 2.0 ** 15, 48000, ll, p1tools.SPREAD_ALPHA) for c in range(2)])
                dv = np.array([p1tools.mapping_from_opus(th[c], 2048, 48000) for c in range(2)])
                g4[f"lv{lv}_f{i}_thres"] = th
                g4[f"lv{lv}_f{i}_div"] = dv
    # other rates / sizes: band-edge table and break-at-empty-band
    for (N, sr) in ((512, 44100), (2048, 96000), (1024, 8000), (640, 32000)):
        xx = pcmformat.to_f64(synth.to_pcm(synth.harmonic_mix(N, 1, sr, seed=N), "s16le"), dt)
        frad, idx, ch, srr = p1.analogue(xx, 16, sr, 1.0)
        raw = zlib.decompress(frad, wbits=-15)
        tl = int.from_bytes(raw[:4], "big")
        g4[f"alt_{N}_{sr}_in"] = synth.to_pcm(synth.harmonic_mix(N, 1, sr, seed=N), "s16le")
        g4[f"alt_{N}_{sr}_q"] = p1tools.exp_golomb_rice_decode(raw[4 + tl:]).astype(np.int32)
        g4[f"alt_{N}_{sr}_tq"] = p1tools.exp_golomb_rice_decode(raw[4:4 + tl]).astype(np.int32)
        g4[f"alt_{N}_{sr}_dec"] = p1.digital(frad, idx, ch, srr, N)
        g4[f"alt_{N}_{sr}_edges"] = np.array([p1tools.get_bin_range(N, sr, b).start for b in range(27)]
                                             + [p1tools.get_bin_range(N, sr, 26).stop], dtype=np.int64)
    # Golomb coder known answers
    gol = []
    for arr in ([0], [1], [-1], [0, 0, 0], [3, -2, 0, 7, -8, 1], list(range(-20, 21)), [1000, -1, 0, 5], []):
        a = np.array(arr, dtype=int)
        gol.append({"data": arr, "hex": p1tools.exp_golomb_rice_encode(a).hex()})
    json.dump(gol, open(os.path.join(OUT, "g4_golomb.json"), "w"))
    g4["hann_128"] = backend.hanning_in_overlap(128)
    np.savez_compressed(os.path.join(OUT, "g4_p1.npz"), **g4)

    # ------------------------------------------------------------------ G5: edge cases
    g5 = {}
    x = np.array([[0.5], [np.nan], [np.inf], [-np.inf], [0.25], [-0.125], [1e-8], [0.0]])
    # NaN compares False in the overflow test; Inf would escalate forever -> keep NaN only for analogue
    xn = np.array([[0.5], [np.nan], [0.25], [-0.125], [1e-8], [0.0], [-0.75], [0.3]])
    for bits in (16, 32, 64):
        frad, idx, ch, sr = p4.analogue(xn, bits, 48000, False)
        g5[f"nan_p4_b{bits}_frad"] = u8(frad); g5[f"nan_p4_b{bits}_dec"] = p4.digital(frad, idx, 1, False)
    g5["nan_in"] = xn
    # a payload holding NaN / +-Inf words decodes to zeros there (profile 0: then IDCT)
    pay = np.array([0.5, np.nan, np.inf, -np.inf, 0.25, -0.125, 1e-8, 0.0]).astype(">f4").tobytes()
    g5["scrub_payload"] = u8(pay)
    g5["scrub_p4_dec"] = p4.digital(pay, 3, 1, False)
    g5["scrub_p0_dec"] = p0.digital(pay, 3, 1, False)
    g5["scrub_p0_dec_c2"] = p0.digital(pay, 3, 2, False)
    # escalation: 1e6 at 16 bit -> 24 bit; 1e39 at 32 -> 48; 70000 at 12 -> 24
    for name, v, bits in (("esc16", 1e6, 16), ("esc32", 1e39, 32), ("esc12", 70000.0, 12), ("esc24", 1e39, 24)):
        xe = np.array([[v], [0.5], [-0.25], [0.125]])
        for prof, mod in ((0, p0), (4, p4)):
            frad, idx, ch, sr = mod.analogue(xe, bits, 48000, False)
            g5[f"{name}_p{prof}_frad"] = u8(frad); g5[f"{name}_p{prof}_idx"] = np.array(idx)
        g5[f"{name}_in"] = xe
    # to_f64 for every format, raw bytes in
    rb = np.random.default_rng(99).integers(0, 256, 64, dtype=np.uint8)
    g5["fmt_bytes"] = rb
    for fmt in ("u8", "u16le", "u16be", "u32le", "u32be", "u64le", "u64be", "s8", "s16le", "s16be", "s32le",
                "s32be", "s64le", "s64be", "f16le", "f16be", "f32le", "f32be", "f64le", "f64be"):
        dtp = pcmformat.ff_format_to_numpy_type(fmt)
        arr = np.frombuffer(rb.tobytes(), dtp)
        conv = pcmformat.to_f64(arr, dtp)
        g5[f"to_f64_{fmt}"] = np.asarray(conv).astype(np.float64) if conv.dtype.kind != "f" else \
            np.asarray(conv).astype(conv.dtype.newbyteorder("="))
        g5[f"to_f64_{fmt}_kind"] = np.array(conv.dtype.str)
    # from_f64 truncation / wrap
    ff = np.array([0.99999, -1.0, 0.5, -0.5, 1.0 / 3, 0.0, 0.999984741, -0.99997])
    for fmt in ("s16le", "s32le", "u8", "u16le", "s8"):
        dtp = pcmformat.ff_format_to_numpy_type(fmt)
        with np.errstate(all="ignore"):
            g5[f"from_f64_{fmt}"] = pcmformat.from_f64(ff, dtp)
    g5["from_f64_in"] = ff
    np.savez_compressed(os.path.join(OUT, "g5_edge.npz"), **g5)

    manifest["files"] = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
