"""Profile 1 (lossy, psychoacoustic quantiser) parity: kernels K7 / K8 and the overlap-add.

Contract: the pre-entropy integers are compared with the oracle / the
reference-generated fixture G4 value by value.  pow / log on the GPU and in numpy may round a
half-way case differently, so the assertion is: |dq| <= 1 everywhere and the fraction of differing
values <= 1e-3 (measured rates are printed); decoded PCM is compared by absolute error and PSNR.
"""
import numpy as np
import pytest

from conftest import load_npz
from helpers import EmuBackend, GpuBackend
from frad_python_amd import synth
from oracle import frad_oracle as fo

_backends = {}


@pytest.fixture(params=[pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    if request.param not in _backends:
        _backends[request.param] = EmuBackend() if request.param == "emu" else GpuBackend()
    return _backends[request.param]


def _check_ints(got, want, what):
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1, f"{what}: max |diff| {d.max()}"
    frac = np.count_nonzero(d) / d.size
    assert frac <= 1e-3, f"{what}: {frac:.2e} of the values differ"
    return frac


def test_p1_golden_g4_quantiser_and_decode(be, g4):
    frames = g4["frames_s16le"]                                # 3 overlapped frames, hop 1920, N = 2048, C = 2
    raw = np.ascontiguousarray(frames)
    for lv in (0, 10, 20):
        ll = 1.25 ** lv / 19.0 + 0.5
        q, tq = be.p1_analogue(raw, "s16le", 3, 2048, 2, 16, 48000, ll)
        for i in range(3):
            wq = np.zeros(4096, np.int32); w = g4[f"lv{lv}_f{i}_q"]; wq[:w.size] = w      # the Golomb decoder drops trailing zeros
            wt = np.zeros(54, np.int32); w = g4[f"lv{lv}_f{i}_tq"]; wt[:w.size] = w
            _check_ints(q[i].reshape(-1), wq, f"q lv{lv} f{i}")
            _check_ints(tq[i].reshape(-1), wt, f"tq lv{lv} f{i}")
            dec = be.p1_digital(wq.reshape(1, 2048, 2), wt.reshape(1, 27, 2), 2048, 2, 16, 48000)[0]
            ref = g4[f"lv{lv}_f{i}_dec"]
            assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("geom", [(2048, 2, 48000, 5), (512, 1, 44100, 3), (2048, 1, 96000, 2), (1024, 2, 8000, 2),
                                  (640, 1, 32000, 2), (128, 2, 48000, 4), (4096, 2, 48000, 1), (2560, 2, 48000, 1),
                                  (2048, 12, 44100, 1), (4096, 6, 48000, 2), (8192, 3, 96000, 1)])   # the last three: channel groups
def test_p1_vs_oracle_sizes_and_rates(be, geom):
    N, C, srate, F = geom
    if be.name == "emu" and N > 2048:
        pytest.skip("emulator: covered by the smaller sizes")
    x = synth.harmonic_mix(F * N, C, srate, seed=N + C)
    raw = synth.to_pcm(x, "s16le")
    dt = fo.pcm_dtype("s16le")
    for loss in (0.553, 5.065):
        q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, srate, loss)
        for f in range(F):
            wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(raw[f * N:(f + 1) * N], dt), 16, srate, loss)
            _check_ints(q[f].reshape(-1), wq, f"q N={N} f{f}")
            _check_ints(tq[f].reshape(-1), wt, f"tq N={N} f{f}")
            dec = be.p1_digital(wq.reshape(1, N, C).astype(np.int32), wt.reshape(1, 27, C).astype(np.int32), N, C, 16, srate)[0]
            ref = fo.p1_digital_post(wq, wt, 2, C, srate, N)
            assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("C", [2, 1])
def test_p1_decode_of_frames_the_tables_do_not_hold(be, C):
    """K8 at N = 2048 is table-driven (band codes 0 .. 255, |q| < 256); anything else -- a corrupt stream's codes, an
    extremely loud bin -- is marked by the wave kernel and decoded again by the exact kernel behind it.  Normal and
    marked frames are mixed (odd frame count: the mono kernel's half-empty last wave)."""
    N, F, srate = 2048, 7, 48000
    rng = np.random.default_rng(77 + C)
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, srate, seed=5 + C), "s16le")
    q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, srate, 0.553)
    q, tq = q.copy(), tq.copy()
    tq[1, 3, C - 1] = 300                                      # beyond the threshold table
    tq[2, 5, 0] = -5                                           # a negative code: threshold < 1 (only a damaged stream has one)
    q[3, 17, 0] = 1000; q[3, 900, C - 1] = -70000              # beyond the |q|^(4/3) table
    tq[5, :, :] = rng.integers(-3, 400, size=(27, C)); q[5, ::7, :] = rng.integers(-500, 500, size=q[5, ::7, :].shape)
    tq[6, 26, 0] = 256                                         # last frame, first code past the table
    dec = be.p1_digital(q, tq, N, C, 16, srate)
    for f in range(F):
        ref = fo.p1_digital_post(q[f].reshape(-1), tq[f].reshape(-1), 2, C, srate, N)
        assert np.all(np.isfinite(ref))
        assert np.max(np.abs(dec[f] - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), f"frame {f}"


@pytest.mark.parametrize("geom", [(640, 2, 32000), (896, 1, 44100), (1536, 2, 48000), (1792, 2, 48000), (3072, 1, 96000), (5120, 2, 44100),
                                  (7168, 1, 48000), (6144, 1, 96000)])
def test_p1_compact_sizes_through_the_mixed_radix_kernels(be, geom):
    """{160, 192, 224} x 2^n (fourier/profiles.py:14-23) in O(N log N): K7 / K8 around the mixed-radix FFT (frad_mixed.hip)."""
    N, C, srate = geom
    if be.name == "emu" and N > 1792:
        pytest.skip("emulator: covered by the smaller sizes")
    F = 2
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, srate, seed=N), "s16le")
    dt = fo.pcm_dtype("s16le")
    for loss in (0.553, 5.065):
        q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, srate, loss)
        for f in range(F):
            wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(raw[f * N:(f + 1) * N], dt), 16, srate, loss)
            _check_ints(q[f].reshape(-1), wq, f"q N={N} f{f}")
            _check_ints(tq[f].reshape(-1), wt, f"tq N={N} f{f}")
            dec = be.p1_digital(wq.reshape(1, N, C).astype(np.int32), wt.reshape(1, 27, C).astype(np.int32), N, C, 16, srate)[0]
            ref = fo.p1_digital_post(wq, wt, 2, C, srate, N)
            assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def test_p1_short_frame_is_zero_padded(be):
    """flush: the last frame is shorter than the compact size and is padded (profile1.py:19)."""
    N, C, nv = 1024, 2, 900
    raw = synth.to_pcm(synth.harmonic_mix(nv, C, 48000, seed=9), "s16le")
    q, tq = be.p1_analogue(raw, "s16le", 1, N, C, 16, 48000, 1.0, n_valid=nv)
    wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(raw, fo.pcm_dtype("s16le")), 16, 48000, 1.0)
    assert aux["dlen"] == N
    _check_ints(q[0].reshape(-1), wq, "q"); _check_ints(tq[0].reshape(-1), wt, "tq")


def test_p1_overlapped_gather_and_overlap_add(be, g4):
    """Encoder overlap = strided frame gather; decoder overlap = Hann cross-fade (R2, R8)."""
    arr = load_npz("g3_p1_streams.npz")
    sig = arr["p1_input_s16le"]                                # 3*1920 + 700 sample-frames, stereo
    q, tq = be.p1_analogue(np.ascontiguousarray(sig), "s16le", 3, 2048, 2, 16, 48000, 0.5 + 1.25 ** 20 / 19.0, frame_stride=1920)
    for i in range(3):
        w = g4[f"lv20_f{i}_q"]; wq = np.zeros(4096, np.int32); wq[:w.size] = w
        _check_ints(q[i].reshape(-1), wq, f"hop q f{i}")
    # decode the reference's own integers, cross-fade on the device, compare with the reference decoder's PCM
    want = arr["p1_lv20_decoded"]
    qs = np.stack([np.pad(g4[f"lv20_f{i}_q"], (0, 4096 - g4[f"lv20_f{i}_q"].size)) for i in range(3)]).reshape(3, 2048, 2)
    ts = np.stack([np.pad(g4[f"lv20_f{i}_tq"], (0, 54 - g4[f"lv20_f{i}_tq"].size)) for i in range(3)]).reshape(3, 27, 2)
    dec = be.p1_digital(qs.astype(np.int32), ts.astype(np.int32), 2048, 2, 16, 48000)
    out, tail = be.p1_ola(dec, 16)
    got = out.reshape(-1, 2)
    assert np.max(np.abs(got - want[:got.shape[0]])) <= 1e-12
    psnr = 10 * np.log10(1.0 / max(np.mean((got - want[:got.shape[0]]) ** 2), 1e-300))
    assert psnr > 200
    # second batch continues from the first one's tail
    out2, tail2 = be.p1_ola(dec[2:3], 16, prev_tail=dec[1, 1920:])
    assert np.max(np.abs(out2[0] - out[2])) == 0.0


@pytest.mark.parametrize("fmt", ["f32le", "f32be", "f16le"])
def test_p1_float_pcm_mixed_precision(be, fmt, g6):
    """f32 / f16 PCM is not widened by the reference (pcmformat.py:35): scipy's DCT then runs in float32, the band
    statistics of p1tools.mask_thres_mos stay float32 and only the divide + quantiser are float64 (profile1.py:21-36).
    The kernels transform in float64 and round the coefficients to float32, so they differ from the reference by its own
    float32 DCT rounding: |dq| <= 1 on at most 2 % of the values (observed rates are printed)."""
    for (N, C, sr) in ((2048, 2, 48000), (640, 1, 32000)):
        raw = g6[f"f_{fmt}_{N}_{C}_in"]
        for lv, ll in (("a", 0.553), ("b", 5.0)):
            q, tq = be.p1_analogue(np.ascontiguousarray(raw), fmt, 1, N, C, 16, sr, ll)
            wq = np.zeros(N * C, np.int64); w = g6[f"f_{fmt}_{N}_{C}_{lv}_q"]; wq[:w.size] = w
            wt = np.zeros(27 * C, np.int64); w = g6[f"f_{fmt}_{N}_{C}_{lv}_tq"]; wt[:w.size] = w
            d = np.abs(q[0].reshape(-1) - wq); dt = np.abs(tq[0].reshape(-1) - wt)
            fq, ft = np.count_nonzero(d) / d.size, np.count_nonzero(dt) / dt.size
            print(f"p1 {fmt} N={N} C={C} loss={ll}: q differs {fq:.2e}, tq differs {ft:.2e}")
            assert d.max() <= 1 and fq <= 2e-2, (fmt, N, C, ll, d.max(), fq)
            assert dt.max() <= 1 and np.count_nonzero(dt) <= 1, (fmt, N, C, ll, dt.max())
    # white noise, against the oracle (which the fixtures above pin on float input)
    rng = np.random.default_rng(5)
    raw = synth.to_pcm(rng.uniform(-0.9, 0.9, (1024, 2)), fmt)
    q, tq = be.p1_analogue(raw, fmt, 1, 1024, 2, 24, 44100, 1.0)
    wq, wt, aux = fo.p1_analogue_pre(np.frombuffer(raw.tobytes(), fo.pcm_dtype(fmt)).reshape(-1, 2), 24, 44100, 1.0)
    assert aux["freqs"].dtype == np.float32
    d = np.abs(q[0].reshape(-1) - wq)
    print(f"p1 {fmt} noise: q differs {np.count_nonzero(d) / d.size:.2e}")
    assert d.max() <= 1 and np.count_nonzero(d) <= 2e-2 * d.size
    assert np.abs(tq[0].reshape(-1) - wt).max() <= 1


@pytest.mark.parametrize("geom", [(10240, 1, 48000), (5120, 2, 44100), (2560, 5, 96000)])
def test_p1_frames_wider_than_a_cu(be, geom, g6):
    """Compact sizes that no LDS-resident kernel can hold (e.g. 10240 mono = 160 KiB of float64 plus tables) run through
    HBM workspaces (csrc/frad_global.hip); they used to be refused.  Reference fixture G6 + oracle."""
    N, C, sr = geom
    if be.name == "emu" and N * C > 10240:
        pytest.skip("emulator: one wide geometry is enough")
    raw = g6[f"w_{N}_{C}_in"]
    q, tq = be.p1_analogue(np.ascontiguousarray(raw), "s16le", 1, N, C, 16, sr, 1.0)
    wq = np.zeros(N * C, np.int64); w = g6[f"w_{N}_{C}_q"]; wq[:w.size] = w
    wt = np.zeros(27 * C, np.int64); w = g6[f"w_{N}_{C}_tq"]; wt[:w.size] = w
    _check_ints(q[0].reshape(-1), wq, f"q {geom}"); _check_ints(tq[0].reshape(-1), wt, f"tq {geom}")
    dec = be.p1_digital(wq.reshape(1, N, C).astype(np.int32), wt.reshape(1, 27, C).astype(np.int32), N, C, 16, sr)[0]
    ref = g6[f"w_{N}_{C}_dec"]
    assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("geom", [(10240, 2, 48000), (12288, 3, 44100), (20480, 1, 96000), (24576, 1, 96000), (28672, 2, 48000)])
def test_p1_widest_compact_sizes_in_n_log_n(be, geom):
    """The compact sizes beyond a CU's LDS (up to 28 672 = 7 x 4096 samples) through the HBM workspaces with the mixed-radix
    FFT (frad_mixed.hip k_gm_*) instead of the dense cosine product; oracle = the reference's arithmetic (scipy)."""
    N, C, srate = geom
    if be.name == "emu" and N > 10240:
        pytest.skip("emulator: one wide geometry is enough")
    F = 2
    raw = synth.to_pcm(synth.harmonic_mix(F * N, C, srate, seed=N + C), "s16le")
    dt = fo.pcm_dtype("s16le")
    q, tq = be.p1_analogue(raw, "s16le", F, N, C, 16, srate, 0.553)
    for f in range(F):
        wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(raw[f * N:(f + 1) * N], dt), 16, srate, 0.553)
        _check_ints(q[f].reshape(-1), wq, f"q N={N} f{f}")
        _check_ints(tq[f].reshape(-1), wt, f"tq N={N} f{f}")
        dec = be.p1_digital(wq.reshape(1, N, C).astype(np.int32), wt.reshape(1, 27, C).astype(np.int32), N, C, 16, srate)[0]
        ref = fo.p1_digital_post(wq, wt, 2, C, srate, N)
        assert np.max(np.abs(dec - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


# ---------------------------------------------------------------------------------------------------------------------
# Exp-Golomb-Rice stage on the device (p1tools.py:46-74, profile1.py:43-45, 59-64): bit-exact, no tolerance
# ---------------------------------------------------------------------------------------------------------------------
def _pad(a, n):
    out = np.zeros(n, np.int32); out[:a.size] = a
    return out


def test_golomb_bodies_equal_the_reference_frames(be, g4, g6):
    """The device coder's pre-deflate body == inflate(reference frame) for G4 (3 loss levels x 3 frames) and the wide
    G6 frames (multi-tile: 10240 / 10240 / 12800 values); decoding the reference's bodies gives its integers back."""
    import zlib
    for lv in (0, 10, 20):
        qs = np.stack([_pad(g4[f"lv{lv}_f{i}_q"], 4096).reshape(2048, 2) for i in range(3)])
        ts = np.stack([_pad(g4[f"lv{lv}_f{i}_tq"], 54).reshape(27, 2) for i in range(3)])
        want = [zlib.decompress(g4[f"lv{lv}_f{i}_frad"].tobytes(), wbits=-15) for i in range(3)]
        assert be.golomb_encode(qs, ts) == want, lv
        q, tq, st = be.golomb_decode(want, 2048, 2)
        assert np.array_equal(q, qs) and np.array_equal(tq, ts) and not st.any(), lv
    for (N, C) in ((10240, 1), (5120, 2), (2560, 5)):
        q = _pad(g6[f"w_{N}_{C}_q"], N * C).reshape(1, N, C); t = _pad(g6[f"w_{N}_{C}_tq"], 27 * C).reshape(1, 27, C)
        want = g6[f"w_{N}_{C}_gol"].tobytes()
        assert be.golomb_encode(q, t) == [want], (N, C)
        dq, dt, st = be.golomb_decode([want], N, C)
        assert np.array_equal(dq, q) and np.array_equal(dt, t), (N, C)


def test_golomb_known_answers_and_long_vectors(be, g6):
    """exp_golomb_rice_encode on single vectors (g4_golomb.json, G6 gol_*): coded as the coefficient stream of a frame
    whose 27 thresholds are zero (that stream is then the 4 bytes 00 ff ff ff e0)."""
    from conftest import load_json
    cases = [(np.array(c["data"], np.int64), bytes.fromhex(c["hex"])) for c in load_json("g4_golomb.json") if c["data"]]
    cases += [(g6[f"gol_{n}_data"], g6[f"gol_{n}_bytes"].tobytes()) for n in ("lap4k", "lap_wide", "sparse", "zeros", "one", "pow2", "big")]
    tq0 = bytes([0]) + bytes([0xff, 0xff, 0xff, 0xe0])
    for data, want in cases:
        n = data.size
        body = be.golomb_encode(data.astype(np.int32).reshape(1, n, 1), np.zeros((1, 27, 1), np.int32))[0]
        assert body[:4] == (5).to_bytes(4, "big") and body[4:9] == tq0, data[:8]
        assert body[9:] == want, (data[:8], body[9:40].hex(), want[:31].hex())
        q, tq, st = be.golomb_decode([body], n, 1)
        assert np.array_equal(q.reshape(-1), data) and not tq.any()


def test_golomb_random_round_trip_and_decoder_end_conditions(be):
    rng = np.random.default_rng(77)
    F, N, C = (3, 640, 2) if be.name == "emu" else (40, 2048, 2)
    q = np.rint(rng.laplace(0, 5.0, (F, N, C)) * rng.integers(0, 2, (F, N, 1))).astype(np.int32)
    q[0] = 0; q[1, 5:] = 0                                     # all-zero frame (k = 0), long zero tail
    tq = rng.integers(0, 40, (F, 27, C)).astype(np.int32)
    bodies = be.golomb_encode(q, tq)
    for f in range(F):
        tg, fg = fo.golomb_encode(tq[f].reshape(-1)), fo.golomb_encode(q[f].reshape(-1))
        assert bodies[f] == len(tg).to_bytes(4, "big") + tg + fg, f
    dq, dt, st = be.golomb_decode(bodies, N, C)
    assert np.array_equal(dq, q) and np.array_equal(dt, tq)
    # the reference decoder's end conditions (p1tools.py:62-74): truncated bodies, a code cut short by the end of the
    # buffer, a body shorter than its length word, a threshold length pointing past the end
    b = bodies[2]
    cut = [b[:len(b) // 2], b[:len(b) - 1], b[:7], b[:3], b"", (1 << 20).to_bytes(4, "big") + b[4:60], b + b"\x00" * 5]
    dq, dt, st = be.golomb_decode(cut, N, C)
    for i, body in enumerate(cut):
        if len(body) < 4:
            assert st[i] == 1 and not dq[i].any() and not dt[i].any()
            continue
        tl = int.from_bytes(body[:4], "big")
        wt, wq = fo.golomb_decode(body[4:4 + tl]) if len(body) > 4 else np.array([]), fo.golomb_decode(body[4 + tl:]) if len(body) > 4 + tl else np.array([])
        wt, wq = np.clip(wt, -2 ** 31, 2 ** 31 - 1)[:27 * C], np.clip(wq, -2 ** 31, 2 ** 31 - 1)[:N * C]
        assert np.array_equal(dt[i].reshape(-1), _pad(wt.astype(np.int64), 27 * C)), i
        assert np.array_equal(dq[i].reshape(-1), _pad(wq.astype(np.int64), N * C)), i


def test_golomb_streams_that_do_not_fall_into_step(be):
    """The wave-per-frame decoder guesses where each lane's chunk of the stream begins and corrects the guesses from the left;
    equal-length codes keep a wrongly started walk out of step for ever, so these streams take the most rounds (and, beyond
    its round limit, the lane-per-frame kernel): k = 3, one 6-bit code, then 4-bit codes only -- every chunk is entered at
    offset 2 (mod 4), never at the guessed 0.  Also a stream whose second half is in step and one of 2-bit codes."""
    N, C = (640, 2) if be.name == "emu" else (2048, 2)
    tq = np.arange(27 * C, dtype=np.int32).reshape(1, 27, C) % 7
    tbytes = fo.golomb_encode(tq.reshape(-1))

    def body_of(k, bits):
        bits = bits + "0" * (-len(bits) % 8)
        return len(tbytes).to_bytes(4, "big") + tbytes + bytes([k]) + int(bits, 2).to_bytes(len(bits) // 8, "big")

    n = N * C
    rng = np.random.default_rng(5)
    tail = "".join("1" + format(int(v), "03b") for v in rng.integers(0, 8, n // 2))
    bodies = [body_of(3, "010000" + "1111" * (n - 1)),
              body_of(3, "010000" + "1111" * (n // 2) + "011" + tail),          # (the walk meets a code cut differently half-way)
              body_of(1, "0100" + "11" * (n - 1)),
              body_of(3, "010000" + "1010" * (n // 3))]                          # ends early: the rest of the frame is zero
    dq, dt, st = be.golomb_decode(bodies, N, C)
    for i, b in enumerate(bodies):
        wq = fo.golomb_decode(b[4 + len(tbytes):])[:n]
        assert np.array_equal(dq[i].reshape(-1), _pad(np.asarray(wq, np.int64), n)), i
        assert np.array_equal(dt[i], tq[0]), i


def test_golomb_fast_kernels_take_what_a_coder_writes(be):
    """Which decoder runs matters: the lane-per-frame kernel costs milliseconds per batch.  The walks (first wave kernel) must settle
    noise-like frames AND tonal ones (the upper bands all zero: a run of one code, where a wrongly started walk never falls into
    step by itself) AND the reference's own frames; very quiet frames (k = 1: '10' repeated reads as a valid code from the wrong
    bit too) may be left to the entry-map kernel behind them, which must take everything a coder can write."""
    rng = np.random.default_rng(2024)
    N, C = 2048, 2
    F = 3 if be.name == "emu" else 24
    noise = np.rint(rng.normal(0, 6, (F, N, C))).astype(np.int32)
    tonal = np.rint(rng.normal(0, 40, (F, N, C))).astype(np.int32); tonal[:, 300:, :] = 0
    gaps = np.rint(rng.normal(0, 9, (F, N, C))).astype(np.int32); gaps[:, 100:900, :] = 0; gaps[:, 1200:, :] = 0
    quiet = (rng.integers(-2, 3, (F, N, C)) * (rng.random((F, N, C)) < 0.2)).astype(np.int32); quiet[:, 0, 0] = 2
    tq = rng.integers(0, 30, (F, 27, C)).astype(np.int32)
    for name, q, walks_only in (("noise", noise, True), ("tonal", tonal, True), ("gaps", gaps, True), ("quiet", quiet, False)):
        bodies = be.golomb_encode(q, tq)
        if walks_only:
            dq, dt, todo = be.golomb_decode_fast_only(bodies, N, C, 0)
            assert not todo.any(), (name, todo)
            assert np.array_equal(dq, q) and np.array_equal(dt, tq), name
        dq, dt, todo = be.golomb_decode_fast_only(bodies, N, C, 1)
        assert not todo.any(), (name, "with maps", todo)
        assert np.array_equal(dq, q) and np.array_equal(dt, tq), name


def test_golomb_decoder_on_damaged_streams(be):
    """Random bit flips and zero runs inside valid bodies: long codes (> 64 bits), spurious early ends, values beyond
    int32 -- the wave-per-frame decoder must hand what it cannot take to the lane-per-frame one, and both must agree with
    the reference decoder's semantics (p1tools.py:62-74; values outside int32 saturate here)."""
    rng = np.random.default_rng(99)
    N, C = (640, 2) if be.name == "emu" else (2048, 2)
    q = np.rint(rng.laplace(0, 8.0, (1, N, C))).astype(np.int32)
    tq = rng.integers(0, 40, (1, 27, C)).astype(np.int32)
    body = bytearray(be.golomb_encode(q, tq)[0])
    tl = int.from_bytes(body[:4], "big")
    variants = []
    for i in range(12 if be.name == "emu" else 60):
        b = bytearray(body)
        kind = i % 4
        at = int(rng.integers(5 + tl + 1, len(b) - 1))
        if kind == 0:
            b[at] ^= 1 << int(rng.integers(0, 8))                                  # one flipped bit
        elif kind == 1:
            n = int(rng.integers(3, 40)); b[at:at + n] = bytes(min(n, len(b) - at))   # a run of zero bytes: one very long code
        elif kind == 2:
            b[4 + tl] = int(rng.integers(0, 40))                                   # another k for the coefficient stream
        else:
            for _ in range(5):
                b[int(rng.integers(5 + tl + 1, len(b)))] = int(rng.integers(0, 256))
        variants.append(bytes(b))
    dq, dt, st = be.golomb_decode(variants, N, C)
    lim = np.iinfo(np.int32)
    for i, b in enumerate(variants):
        wq = fo.golomb_decode(b[4 + tl:])[:N * C]
        wq = np.array([max(lim.min, min(lim.max, int(v))) for v in wq], dtype=np.int64)
        assert np.array_equal(dq[i].reshape(-1), _pad(wq, N * C)), (i, i % 4)
        assert np.array_equal(dt[i], tq[0]), i
