"""Decoder output conversion (from_f64, backend/pcmformat.py:49-62 + src/decoder.py:23) and the native ASFH scan
(tools/asfh.py:98-134, decoder.py:82-106).  from_f64 is bit-exact against reference-generated fixtures (G5, G6); the
out-of-range behaviour is pinned against numpy's astype on this (x86-64) host, which is what the reference runs on."""
import numpy as np
import pytest

from conftest import load_json, load_npz
from helpers import EmuBackend, GpuBackend
from frad_python_amd import synth
from oracle import frad_oracle as fo

_backends = {}
INT_FORMATS = ("u8", "u16le", "u16be", "u32le", "u32be", "s8", "s16le", "s16be", "s32le", "s32be", "s64le", "s64be", "u64le")


@pytest.fixture(params=[pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    if request.param not in _backends:
        _backends[request.param] = EmuBackend() if request.param == "emu" else GpuBackend()
    return _backends[request.param]


def test_from_f64_reference_fixtures(be, g5, g6):
    ff = g6["ff_in"]
    for fmt in INT_FORMATS:
        got = be.from_f64(ff, fmt)
        # the fixture is from_f64's own return value: for big-endian formats that is the float64 input, unscaled (the
        # reference compares against native dtypes only); its caller then applies .astype(fmt) (src/decoder.py:23)
        ref = np.frombuffer(g6[f"ff_{fmt}"].tobytes(), np.dtype(str(g6[f"ff_{fmt}_dtype"])))
        with np.errstate(all="ignore"):
            assert got.tobytes() == ref.astype(fo.pcm_dtype(fmt)).tobytes(), fmt
    x5 = g5["from_f64_in"]
    for fmt in ("s16le", "s32le", "u8", "u16le", "s8"):
        assert np.array_equal(be.from_f64(x5, fmt), g5[f"from_f64_{fmt}"]), fmt
    # floats: astype (round to nearest even), both byte orders
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-1, 1, 1000), [0.0, -0.0, 1e-8, 65504.0, 65520.0, 1e-40, 3.5e38, np.inf, -np.inf]])
    for fmt in ("f16le", "f16be", "f32le", "f32be", "f64le", "f64be"):
        with np.errstate(all="ignore"):
            want = x.astype(fo.pcm_dtype(fmt))
        assert be.from_f64(x, fmt).tobytes() == want.tobytes(), fmt


def test_from_f64_out_of_range_like_numpy_on_x86(be):
    """A lossy decode overshoots +-1 now and then; the reference then gets whatever numpy's float -> int astype yields on
    its host (x86-64: cvttsd2si semantics).  Ragged length and odd alignment on the way."""
    x = np.array([1.0, 1.2, -1.0001, -1.5, 3.0, 70000.0, -70000.0, 1e10, -1e10, 1e20, -1e20, np.nan, np.inf, -np.inf,
                  0.999999, -0.3, 0.25])
    for fmt in INT_FORMATS:
        if fmt.startswith(("u32", "u64")):
            # numpy 2.2's own answer for float64 -> uint32 / uint64 overflow depends on the loop it picks (the contiguous
            # SIMD loop gives 0 where the scalar / strided loop wraps): there is nothing to pin beyond the in-range fixtures
            continue
        with np.errstate(all="ignore"):
            want = fo.from_f64(x, fo.pcm_dtype(fmt)).astype(fo.pcm_dtype(fmt))
        assert be.from_f64(x, fmt).tobytes() == want.tobytes(), fmt


@pytest.mark.parametrize("C", [2, 1])
@pytest.mark.parametrize("fmt", ["s16le", "s32be", "f32le", "u8", "s24" if False else "u16le"])
def test_digital_with_output_format(be, fmt, C):
    """frad_p4_digital_pcm (fused) / frad_p0_digital_pcm == from_f64(digital(...)).astype(fmt), bit for bit."""
    # N = 2048 with 2 channels or 1: the conversion is fused into the wave kernel's output stage (mono: two frames per
    # wave, so an odd frame count leaves a half-empty last unit)
    N, F = (256, 3) if be.name == "emu" and fmt not in ("s16le", "f32le") else (2048, 3) if be.name == "emu" else (2048, 9)
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, 48000, seed=4), "s16le")
    dt = fo.pcm_dtype(fmt)
    for profile in (4, 0):
        for bits, le in ((16, False), (32, True), (12, False), (64, False)):
            frames = [(fo.p4_analogue if profile == 4 else fo.p0_analogue)(fo.to_f64(raw[f * N:(f + 1) * N], fo.pcm_dtype("s16le")), bits, 48000, le)
                      for f in range(F)]
            pay = np.stack([np.frombuffer(fr[0], np.uint8) for fr in frames])
            got = be.digital_pcm(profile, pay, F, N, C, bits, le, fmt)
            f64 = be.digital(profile, pay, F, N, C, bits, le)
            with np.errstate(all="ignore"):
                want = fo.from_f64(f64, dt).astype(dt)
            assert got.tobytes() == want.tobytes(), (profile, bits, le, fmt)
            if profile == 4:                                   # and against the oracle end to end (profile 4 is bit-exact)
                ref = np.stack([fo.p4_digital(fr[0], fr[1], C, le) for fr in frames])
                with np.errstate(all="ignore"):
                    assert got.tobytes() == fo.from_f64(ref, dt).astype(dt).tobytes()


def _scan_all(lib, data, chunked=None):
    rows, pos, why = lib.asfh_scan(data)
    return rows, pos, why


@pytest.mark.parametrize("geom", [(1024, 2), (4096, 2), (896, 2), (300, 3), (512, 8), (4096, 8), (2560, 1)])
@pytest.mark.parametrize("fmt", ["s16le", "f32le", "s32be"])
def test_decode_with_conversion_in_the_kernels_own_store(be, geom, fmt):
    """frad_p0_digital_pcm / frad_p1_digital_pcm outside the N = 2048 wave kernels: the one-shot, channel-group, mixed-radix,
    Bluestein and direct kernels convert in their own store (one pass, no scratch) == from_f64(digital(...)).astype(fmt)."""
    N, C = geom
    if be.name == "emu" and N * C > 4096:
        pytest.skip("emulator: covered by the smaller geometries")
    F = 3
    dt = fo.pcm_dtype(fmt)
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, 48000, seed=N + C), "s16le")
    for bits, le in ((32, False), (16, True)):
        frames = [fo.p0_analogue(fo.to_f64(raw[f * N:(f + 1) * N], fo.pcm_dtype("s16le")), bits, 48000, le) for f in range(F)]
        pay = np.stack([np.frombuffer(fr[0], np.uint8) for fr in frames])
        got = be.digital_pcm(0, pay, F, N, C, bits, le, fmt)
        f64 = be.digital(0, pay, F, N, C, bits, le)
        with np.errstate(all="ignore"):
            want = fo.from_f64(f64, dt).astype(dt)
        assert got.tobytes() == want.tobytes(), (N, C, bits, fmt)
    # profile 1 at the compact sizes among them
    if N in (1024, 4096, 896, 512, 2560) and C <= 2:
        q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, 48000, 0.553)
        got = be.p1_digital_pcm(q, tq, N, C, 16, 48000, fmt)
        f64 = be.p1_digital(q, tq, N, C, 16, 48000)
        with np.errstate(all="ignore"):
            want = fo.from_f64(f64, dt).astype(dt)
        assert got.tobytes() == want.tobytes(), ("p1", N, C, fmt)


def test_which_geometries_still_take_a_second_pass(be):
    """frad_p0_digital_pcm / frad_p1_digital_pcm convert in the transform kernel's own store -- no float64 scratch, no
    second pass -- at N = 1024 (unit kernel) and N = 4096 (two-pass whole-row kernel) as everywhere else below the CU-wide frames; what
    is left: the N = 2048 wave kernel's geometries with an output format other than s16le / s32le / f32le, profile 1 at N = 2048
    (K8 writes float64 for the overlap-add, whose own store converts) and frames wider than a CU.  The library counts its
    second passes (frad_debug_second_passes, not part of the ABI)."""
    import ctypes
    if be.name == "emu":
        dll = be.lib.dll
    else:
        from frad_python_amd import _lib
        dll = _lib.load().dll
    fn = dll.frad_debug_second_passes
    fn.restype = ctypes.c_longlong
    F = 2
    big = be.name != "emu"
    cases = [((1024, 2), "s16le", 32, 0), ((1024, 1), "u8", 16, 0), ((1024, 2), "f64be", 24, 0), ((2048, 2), "s16le", 16, 0),
             ((2048, 2), "u8", 16, 1), ((2048, 2), "s16be", 24, 0), ((896, 2), "s16le", 16, 0), ((300, 3), "s32be", 32, 0)]
    if big:
        cases += [((4096, 2), "s16le", 32, 0), ((4096, 8), "s16le", 32, 0), ((4096, 8), "f32le", 16, 0), ((8192, 2), "s16le", 32, 0),
                  ((28672, 2), "s16le", 32, 1)]
    for (N, C), fmt, bits, second in cases:
        raw = synth.to_pcm(synth.harmonic_mix(F * N, C, 48000, seed=N + C), "s16le")
        frames = [fo.p0_analogue(fo.to_f64(raw[f * N:(f + 1) * N], fo.pcm_dtype("s16le")), bits, 48000, False) for f in range(F)]
        pay = np.stack([np.frombuffer(fr[0], np.uint8) for fr in frames])
        before = fn()
        got = be.digital_pcm(0, pay, F, N, C, bits, False, fmt)
        assert fn() - before == second, (N, C, fmt, bits)
        dt = fo.pcm_dtype(fmt)
        with np.errstate(all="ignore"):
            want = fo.from_f64(be.digital(0, pay, F, N, C, bits, False), dt).astype(dt)
        assert got.tobytes() == want.tobytes(), (N, C, fmt, bits)
    # profile 1: N = 2048 keeps the second pass (K8 -> float64), the other compact sizes do not
    for N, C, second in ((2048, 2, 1), (1024, 2, 0), (512, 1, 0)) + (((4096, 2, 0),) if big else ()):
        raw = synth.to_pcm(synth.harmonic_mix(F * N, C, 48000, seed=N), "s16le")
        q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, 48000, 0.553)
        before = fn()
        be.p1_digital_pcm(q, tq, N, C, 16, 48000, "s16le")
        assert fn() - before == second, ("p1", N, C)


@pytest.mark.parametrize("fmt", ["s16le", "s24le" if False else "s32le", "f32be", "u8", "s16be"])
def test_overlap_add_with_output_format(be, fmt):
    """frad_p1_overlap_add_pcm == from_f64(frad_p1_overlap_add(...)).astype(fmt) (decoder.py:28-46 then src/decoder.py:23), tail in float64"""
    rng = np.random.default_rng(5)
    F, N, C, ratio = 5, 640, 2, 16
    frames = rng.uniform(-1.1, 1.1, (F, N, C))
    prev = rng.uniform(-1, 1, (N - N * (ratio - 1) // ratio, C))
    dt = fo.pcm_dtype(fmt)
    for pt in (None, prev):
        want, wtail = be.p1_ola(frames, ratio, pt)
        got, gtail = be.p1_ola_pcm(frames, ratio, fmt, pt)
        with np.errstate(all="ignore"):
            ref = fo.from_f64(want, dt).astype(dt)
        assert got.tobytes() == ref.tobytes(), fmt
        assert np.array_equal(gtail, wtail)


def test_native_header_scan_equals_the_reference_parser():
    """frad_asfh_scan (host C++, no device) against the oracle's restatement of ASFH.read on reference-generated streams
    (G3: lossless incl. ECC-free variants, profile 1 with force-flush headers), with garbage in front, a split signature
    and truncated tails."""
    from frad_python_amd._lib import FradLib
    from helpers import build_emulator
    lib = FradLib(build_emulator())                            # the scan is host code: same source in both builds
    g3 = load_json("g3_streams.json"); arr = load_npz("g3_p1_streams.npz")
    from test_stream import _inputs
    inputs = _inputs()
    streams = []
    for c in g3["cases"]:
        p = c["params"]
        streams.append(arr[f"{c['name']}_stream"].tobytes() if p["profile"] == 1 else fo.encode_stream(inputs[c["name"].split("_")[0]], **p))
    for s in streams:
        data = b"junk\xff\xd0\xd2" + s                           # garbage with a near-signature in front
        rows, pos, why = lib.asfh_scan(data)
        want, at = [], data.find(b"\xff\xd0\xd2\x98")
        while at < len(data):
            f, hlen = fo.asfh_parse(data, at)
            want.append((at, f, hlen))
            at += hlen + (0 if f["force_flush"] else f["frmbytes"])
        assert len(rows) == len(want) and why == 0 and pos >= len(data) - 3
        for r, (at, f, hlen) in zip(rows, want):
            assert r["header_off"] == at and r["payload_off"] == at + hlen and bool(r["force_flush"]) == bool(f["force_flush"])
            assert (r["profile"], r["channels"], r["srate"]) == (f["profile"], f["channels"], f["srate"])
            if not f["force_flush"]:
                assert r["payload_bytes"] == f["frmbytes"] and r["fsize"] == f["fsize"] and r["depth_idx"] == f["depth_idx"]
                assert bool(r["little_endian"]) == bool(f["little_endian"]) and r["overlap_ratio"] == f["overlap_ratio"]
        # truncation: inside the last payload, inside a header, inside a signature
        last = [w for w in want if not w[1]["force_flush"]][-1]
        for cut, reason in ((last[0] + last[2] + 5, 2), (last[0] + 6, 1), (last[0] + 2, 0)):
            rows2, pos2, why2 = lib.asfh_scan(data[:cut])
            assert why2 == reason and pos2 <= last[0] and all(r["header_off"] < last[0] for r in rows2), (cut, reason)
            # resuming with the rest appended finds the same frames
            rows3, pos3, why3 = lib.asfh_scan(data, start=pos2)
            assert len(rows2) + len(rows3) == len(rows) and why3 == 0
