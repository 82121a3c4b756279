"""Randomised differential test on the MI355X: profile 0 / 4 over random geometry (frame length incl. odd
sizes, channels, PCM format, storage depth, endianness, buffer offsets) against the oracle, under the same
tolerance contract as tests/test_parity.py.  Fixed seed.  Nothing legal is refused: frames wider than a CU's LDS run
through HBM workspaces (csrc/frad_global.hip)."""
import os

import numpy as np
import pytest

from frad_python_amd import synth
from oracle import frad_oracle as fo
from helpers import GpuBackend, oracle_frames, payload_values
from test_parity import EPS32, EPS64

pytestmark = pytest.mark.gpu
FORMATS = ["s16le", "s16be", "u8", "s8", "s32le", "u16le", "f32le", "f32be", "f64le", "f64be", "f16le", "s64le", "u32be"]


def _check(be, rng, profile, N, C, F, fmt, bits, le, offset, hop=None, pad=0, raw_be=True):
    hop = N if hop is None else hop
    raw = synth.to_pcm(rng.uniform(-1, 1, ((F - 1) * hop + N, C)), fmt)
    pay, am = be.analogue(profile, raw, fmt, F, N, C, bits, le, offset=offset, frame_stride=hop, pad_stride=pad, raw_be=raw_be)
    ref = oracle_frames(fo, profile, raw, fmt, F, N, C, bits, le, frame_stride=hop, raw_be=raw_be)
    f32 = fmt.startswith(("f32", "f16"))
    lg = max(np.log2(N), 1.0)
    # 12-bit payloads are always big-endian (profile0.py:29-30); the helper mirrors that
    for f in range(F):
        if profile == 4:
            assert np.array_equal(pay[f], ref[f][0]), ("p4 payload", N, C, F, fmt, bits, le, offset, f)
            continue
        gv, wv = payload_values(fo, pay[f], bits, le), payload_values(fo, ref[f][0], bits, le)
        store = {12: 2.0 ** -7, 16: 2.0 ** -10, 24: 2.0 ** -15, 32: 2.0 ** -23, 48: 2.0 ** -36, 64: 0.0}[bits]
        # unscaled big-endian ints (the reference's quirk) overflow the small storage floats: +-inf must match exactly
        fin = np.isfinite(wv)
        assert np.array_equal(gv[~fin], wv[~fin]), ("p0 non-finite", N, C, F, fmt, bits, le, offset, f)
        if not fin.any():
            continue
        tol = (store + (8 * EPS32 if f32 else 8 * EPS64) * lg) * max(np.max(np.abs(wv[fin])), 1e-300)
        assert np.max(np.abs(gv[fin] - wv[fin])) <= 2 * tol, ("p0 payload", N, C, F, fmt, bits, le, offset, f)
    dec = be.digital(profile, np.stack([r[0] for r in ref]), F, N, C, bits, le, offset=offset)
    for f in range(F):
        if profile == 4:
            assert np.array_equal(dec[f], ref[f][1]), ("p4 decode", N, C, F, fmt, bits, le, offset, f)
        else:
            scale = max(1.0, float(np.max(np.abs(ref[f][1]))))
            assert np.all(np.isfinite(dec[f])) and np.max(np.abs(dec[f] - ref[f][1])) <= 16 * EPS64 * lg * scale, ("p0 decode", N, C, F, fmt, bits, le, offset, f)
    return "ok"


def test_random_geometries_against_oracle():
    be = GpuBackend()
    rng = np.random.default_rng(int(os.environ.get("FRAD_FUZZ_SEED", "20261004")))
    done = {"ok": 0}
    pow2 = [128, 256, 512, 1024, 2048, 4096, 8192, 16384]
    rounds = int(os.environ.get("FRAD_FUZZ_N", "220"))       # a longer one-off hunt: FRAD_FUZZ_N=3000 FRAD_FUZZ_SEED=...
    for i in range(rounds):
        profile = int(rng.choice([0, 0, 4]))
        kind = rng.integers(0, 3) if i % 20 else 3          # kind 3: an odd frame wider than a CU's LDS (a long clip's tail)
        N = int(rng.choice(pow2)) if kind == 0 else int(rng.integers(1, 3000)) if kind == 1 else \
            int(rng.integers(4097, 16384)) if kind == 3 else int(rng.choice([896, 1920, 441, 1000, 1536, 2047, 2049, 96, 95, 97]))
        C = int(rng.choice([1, 1, 2, 2, 2, 3, 5, 6, 8]))
        F = int(rng.choice([1, 2, 3, 7, 33]))
        if kind == 3:
            F, C = int(rng.choice([1, 2])), int(rng.choice([1, 2, 3]))
        if i % 10 == 5:                                      # frames of exactly two channel groups (the two-pass kernels: a frame's
            N, C = [(2048, 16), (4096, 8), (8192, 4), (16384, 2)][int(rng.integers(0, 4))]   # rows split between two blocks,
            F = int(rng.choice([1, 3, 9, 17]))               # partners 8 blocks apart -- frame counts around that guard)
        while N * C * F > 600000:
            F = max(1, F // 2)
            if F == 1 and N * C > 600000:
                C = 1
        fmt = str(rng.choice(FORMATS))
        bits = int(rng.choice(fo.DEPTHS))
        le = bool(rng.integers(0, 2))
        offset = int(rng.choice([0, 0, 0, 2, 6]))
        hop = N if rng.integers(0, 4) else max(1, N - int(rng.integers(0, max(1, N // 8) + 1)))    # overlap read (encoder.py:35-51)
        pad = int(rng.choice([0, 0, 0, 16, 5]))
        raw_be = bool(rng.integers(0, 4))                    # False: big-endian ints normalised like little-endian ones
        done[_check(be, rng, profile, N, C, F, fmt, bits, le, offset, hop, pad, raw_be)] += 1
        if (i + 1) % 5000 == 0:
            print("fuzz progress:", i + 1, done, flush=True)
    print("fuzz:", rounds, "rounds", done)
    assert done["ok"] == rounds, done


def test_random_profile1_geometries_against_oracle():
    """Profile 1 (K7 / K8) over random legal compact frame sizes, rates, depths, loss levels and channel counts."""
    from frad_python_amd.fourier import profiles
    from test_parity_p1 import _check_ints
    be = GpuBackend()
    rng = np.random.default_rng(int(os.environ.get("FRAD_FUZZ_SEED", "20261004")) + 1)
    rounds = int(os.environ.get("FRAD_FUZZ_N", "220")) // 4
    sizes = list(profiles.compact.SAMPLES)                  # every legal compact frame size (profiles.py:14-23), up to 28 672
    dt = fo.pcm_dtype("s16le")
    ok = 0
    for _ in range(rounds):
        N = int(rng.choice(sizes)); C = int(rng.choice([1, 2, 2, 3, 6])); F = int(rng.choice([1, 2, 5]))
        srate = int(rng.choice(profiles.compact.SRATES)); bits = int(rng.choice([8, 12, 16, 24, 32]))
        loss = float(1.25 ** int(rng.integers(0, 21)) / 19.0 + 0.5)
        raw = synth.to_pcm(synth.harmonic_mix(F * N, C, srate, seed=int(rng.integers(0, 1 << 30))) * rng.uniform(0.05, 1.0), "s16le")
        q, tq = be.p1_analogue(raw, "s16le", F, N, C, bits, srate, loss)        # every legal geometry runs (refused == 0)
        for f in range(F):
            wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(raw[f * N:(f + 1) * N], dt), bits, srate, loss)
            _check_ints(q[f].reshape(-1), wq, f"q {(N, C, srate, bits, loss, f)}")
            _check_ints(tq[f].reshape(-1), wt, f"tq {(N, C, srate, bits, loss, f)}")
            dec = be.p1_digital(wq.reshape(1, N, C).astype(np.int32), wt.reshape(1, 27, C).astype(np.int32), N, C, bits, srate)[0]
            ref = fo.p1_digital_post(wq, wt, fo.P1_DEPTHS.index(bits), C, srate, N)
            assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), (N, C, srate, bits, loss, f)
        ok += 1
    print("p1 fuzz:", rounds, "rounds", {"ok": ok, "refused": 0})
    assert ok == rounds


def test_random_entropy_stage_and_output_conversion():
    """Round-2 entry points under random geometry: Exp-Golomb-Rice coder / decoder against the oracle's coder (any legal
    compact size, 1-6 channels, value ranges from all-zero to int32 extremes) and the decode-with-conversion path against
    from_f64 of the float64 decode (every PCM format, N = 2048 stereo fused kernel included)."""
    be = GpuBackend()
    rng = np.random.default_rng(int(os.environ.get("FRAD_FUZZ_SEED", "20261004")) + 2)
    rounds = max(10, int(os.environ.get("FRAD_FUZZ_N", "220")) // 8)
    from frad_python_amd.fourier import profiles
    fmts = ["u8", "u16le", "u16be", "s8", "s16le", "s16be", "s32le", "s32be", "s64le", "f16le", "f32le", "f32be", "f64be"]
    for i in range(rounds):
        N = int(rng.choice(profiles.compact.SAMPLES[:20])); C = int(rng.choice([1, 2, 2, 3, 6])); F = int(rng.choice([1, 3, 9]))
        spread = float(rng.choice([0.0, 0.7, 5.0, 300.0, 2e5]))
        q = np.rint(rng.laplace(0, spread, (F, N, C)) if spread else np.zeros((F, N, C))).clip(-2 ** 31 + 1, 2 ** 31 - 1).astype(np.int32)
        if rng.integers(0, 6) == 0:
            q[0, int(rng.integers(0, N)), 0] = int(rng.choice([2 ** 31 - 1, -2 ** 31 + 1, 2 ** 30, -2 ** 30 - 1]))
        tq = rng.integers(0, 60, (F, 27, C)).astype(np.int32)
        bodies = be.golomb_encode(q, tq)
        for f in range(F):
            tg, fg = fo.golomb_encode(tq[f].reshape(-1)), fo.golomb_encode(q[f].reshape(-1))
            assert bodies[f] == len(tg).to_bytes(4, "big") + tg + fg, ("golomb encode", N, C, F, spread, f)
        dq, dt, st = be.golomb_decode(bodies, N, C)
        assert np.array_equal(dq, q) and np.array_equal(dt, tq) and not st.any(), ("golomb decode", N, C, F, spread)
        # decode with the output conversion
        Np = int(rng.choice([2048, 2048, 1024, 896, 4096, 300])); Cp = int(rng.choice([1, 2, 2, 4])); Fp = int(rng.choice([1, 4]))
        bits = int(rng.choice([16, 32, 64, 24])); le = bool(rng.integers(0, 2)); fmt = str(rng.choice(fmts)); prof = int(rng.choice([0, 0, 4]))
        x = rng.uniform(-1.2, 1.2, (Fp, Np * Cp))
        pay = np.stack([np.frombuffer(fo.pack_floats(x[f], bits, le), np.uint8) for f in range(Fp)])
        got = be.digital_pcm(prof, pay, Fp, Np, Cp, bits, le, fmt)
        f64 = be.digital(prof, pay, Fp, Np, Cp, bits, le)
        dt_ = fo.pcm_dtype(fmt)
        if fmt.startswith(("u32", "u64")):
            continue
        with np.errstate(all="ignore"):
            want = fo.from_f64(f64, dt_).astype(dt_)
        assert got.tobytes() == want.tobytes(), ("digital_pcm", prof, Np, Cp, Fp, bits, le, fmt)
    print("entropy / epilogue fuzz:", rounds, "rounds ok")
