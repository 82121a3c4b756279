"""Drop-in boundary: the reference's streaming API (Encoder.process/flush, Decoder.process/flush)
reproduced byte-for-byte.  Golden streams come from the reference itself (tests/golden/g3_*).

CPU leg ("host"): the product Encoder/Decoder with the TEST bridge (oracle arithmetic) -- checks the
host logic only.  GPU leg: the same objects on the HIP transform core (the product path)."""
import hashlib

import numpy as np
import pytest

from conftest import load_json, load_npz
from frad_python_amd import Decoder, Encoder, synth
from oracle import frad_oracle as fo


def _bridge(kind):
    if kind == "host":
        from helpers import OracleBridge
        return OracleBridge()
    if kind == "host-scan":                                   # oracle arithmetic + the native header scanner (host C++)
        from helpers import OracleBridge, build_emulator
        from frad_python_amd._lib import FradLib
        b = OracleBridge()
        b.scan_lib = FradLib(build_emulator())
        return b
    from frad_python_amd.bridge import HipBridge
    return HipBridge()


@pytest.fixture(params=[pytest.param("host"), pytest.param("host-scan"), pytest.param("gpu", marks=pytest.mark.gpu)])
def kind(request):
    return request.param


def _inputs():
    return {
        "cfg1": synth.sine(48000, 1, 48000, 440.0, 0.5).astype(">f8").tobytes(),
        "tiny": np.array([0.25, -0.5, 0.75, 0.125]).astype(">f8").tobytes(),
        "st": synth.to_pcm(synth.harmonic_mix(3000, 2, 44100, seed=3), "s16le").tobytes(),
        "p1": load_npz("g3_p1_streams.npz")["p1_input_s16le"].tobytes(),
    }


def _encode(kind, pcm, chunk, p):
    enc = Encoder(p["profile"], p["srate"], p["channels"], p["bits"], p["frame_size"], p["pcm_format"], bridge=_bridge(kind))
    enc.set_little_endian(p.get("little_endian", False))
    enc.set_overlap_ratio(p.get("overlap_ratio", 0))
    enc.set_loss_level(p.get("loss_level", 0.5))
    out, samples = b"", 0
    for i in range(0, len(pcm), chunk):
        r = enc.process(pcm[i:i + chunk]); out += r.buf; samples += r.samples
    r = enc.flush(); out += r.buf; samples += r.samples
    return out, samples


def _decode(kind, stream, chunk, channels):
    dec = Decoder(bridge=_bridge(kind))
    pcm, frames = [], 0
    for i in range(0, len(stream), chunk):
        d = dec.process(stream[i:i + chunk]); pcm.append(d.pcm.reshape(-1, channels)); frames += d.frames
    pcm.append(dec.flush().pcm.reshape(-1, channels))
    return np.concatenate(pcm), frames


def _lossless_words(stream: bytes, bits: int) -> np.ndarray:
    """the stored words of every frame of a lossless stream (headers and their CRCs left out; asfh.py:98-134)"""
    out, pos = [], 0
    while pos < len(stream):
        f, hlen = fo.asfh_parse(stream, pos)
        pos += hlen
        if f["force_flush"]:
            continue
        out.append(np.frombuffer(stream[pos:pos + f["frmbytes"]], np.dtype(">u%d" % (bits // 8))))
        pos += f["frmbytes"]
    return np.concatenate(out)


def _p1_frames(stream: bytes):
    """[(header fields, tq ints, q ints)] of a profile-1 stream (ref: profile1.py:43-50, 59-64; asfh.py:98-134)"""
    import struct
    import zlib
    out, pos = [], 0
    while pos < len(stream):
        f, hlen = fo.asfh_parse(stream, pos)
        pos += hlen
        if f["force_flush"]:
            continue
        body = zlib.decompress(stream[pos:pos + f["frmbytes"]], wbits=-15)
        pos += f["frmbytes"]
        tlen = struct.unpack(">I", body[:4])[0]
        head = tuple(f[k] for k in ("profile", "depth_idx", "channels", "srate", "fsize", "overlap_ratio"))
        out.append((head, fo.golomb_decode(body[4:4 + tlen]), fo.golomb_decode(body[4 + tlen:])))
    return out


def test_streams_encode_byte_for_byte_and_decode(kind):
    g3, inputs = load_json("g3_streams.json"), _inputs()
    arr = load_npz("g3_p1_streams.npz")
    for c in g3["cases"]:
        p = c["params"]
        pcm = inputs[c["name"].split("_")[0]]
        lossy = p["profile"] == 1
        for chunk in ((32768, 1000) if not lossy else (4096,)):
            out, samples = _encode(kind, pcm, chunk, p)
            assert samples == c["samples"], c["name"]
            assert len(out) == c["nbytes"] or (kind == "gpu" and lossy), c["name"]
            if kind.startswith("host") or p["profile"] == 4:
                assert hashlib.sha256(out).hexdigest() == c["sha256"], (c["name"], chunk)
            elif not lossy:
                # profile 0 on the GPU: payload words identical up to rare double-rounding ties -> compare values
                ref = fo.encode_stream(pcm, **p)
                a, b = np.frombuffer(out, np.uint8), np.frombuffer(ref, np.uint8)
                assert a.size == b.size, c["name"]
                if p["bits"] <= 32 and p["bits"] % 16 == 0:       # stored words bit-identical up to rare rounding ties: the word
                    wa, wb = _lossless_words(out, p["bits"]), _lossless_words(ref, p["bits"])    # contract of DESIGN.md section 5
                    nd = int(np.count_nonzero(wa != wb))
                    print(f"[p0 gpu stream] {c['name']} chunk {chunk}: {nd} of {wa.size} stored words differ from the reference's")
                    assert nd <= max(2, 1e-5 * wa.size), c["name"]
                elif p["bits"] <= 32:
                    assert np.count_nonzero(a != b) <= max(16, a.size * 1e-4), c["name"]
                # any depth: what the stream decodes to (through the oracle) is the reference's PCM
                assert np.max(np.abs(fo.decode_stream(out) - fo.decode_stream(ref))) <= 1e-12, c["name"]
            else:
                # profile 1 on the GPU: Encoder -> HIP q/tq -> host Golomb + deflate -> ASFH.  The integers may differ
                # from the reference's by one at a rounding tie (contract: |dq| <= 1, <= 1e-3 of them), so compare what
                # the streams hold: frame by frame the same headers, the same integer arrays up to that contract, and
                # the same decoded PCM (through the oracle) as the reference stream.
                ref = arr[f"{c['name']}_stream"].tobytes()
                fa, fb = _p1_frames(out), _p1_frames(ref)
                assert len(fa) == len(fb) == c["frames"], c["name"]
                nq = dq = 0
                for (ha, tqa, qa), (hb, tqb, qb) in zip(fa, fb):
                    assert ha == hb, c["name"]                 # profile, depth index, channels, srate, fsize, overlap
                    n = max(len(qa), len(qb))
                    qa, qb = np.pad(qa, (0, n - len(qa))), np.pad(qb, (0, n - len(qb)))     # the coder drops trailing zeros
                    assert np.max(np.abs(qa - qb)) <= 1 and np.array_equal(tqa[:len(tqb)], tqb[:len(tqa)]), c["name"]
                    nq += n; dq += int(np.count_nonzero(qa != qb))
                assert dq <= 1e-3 * nq, (c["name"], dq, nq)
                print(f"[p1 gpu stream] {c['name']}: {dq} of {nq} quantised bins differ from the reference's, "
                      f"{len(out)} vs {len(ref)} bytes, identical={out == ref}")
                a, b = fo.decode_stream(out), fo.decode_stream(ref)
                assert a.shape == b.shape, c["name"]
                err = np.max(np.abs(a - b))
                psnr = 10 * np.log10(1.0 / max(np.mean((a - b) ** 2), 1e-300))
                assert (dq == 0 and err <= 1e-9) or psnr > 100, (c["name"], err, psnr)
        # decode the REFERENCE stream
        ref_stream = arr[f"{c['name']}_stream"].tobytes() if lossy else fo.encode_stream(pcm, **p)
        got, frames = _decode(kind, ref_stream, 777, p["channels"])
        want = fo.decode_stream(ref_stream)
        assert frames == c["frames"] and list(got.shape) == c["decoded_shape"], c["name"]
        if kind.startswith("host") or p["profile"] == 4:
            assert hashlib.sha256(np.ascontiguousarray(got).astype("<f8").tobytes()).hexdigest() == c["decoded_sha256"], c["name"]
        else:
            assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want))), c["name"]


def test_encoder_rejects_like_the_reference(kind):
    # bits = 0 (the CLI default) fails verify_bit_depth: the reference's constructor ignores the error string
    # and the encoder then silently emits nothing (encoder.py:33, 57-58, 139-140)
    enc = Encoder(4, 48000, 2, 0, 2048, "s16le", bridge=_bridge(kind))
    assert enc.process(b"\x00" * 65536).buf == b"" and enc.flush().buf == b""
    assert isinstance(Encoder(0, 48000, 2, 16, 2048, "s16le", bridge=_bridge(kind)).set_bit_depth(13), str)
    with pytest.raises(SystemExit):
        Encoder(2, 48000, 2, 16, 2048, "s16le")
    with pytest.raises(SystemExit):
        Encoder(0, 48000, 2, 16, 2048, "s17le")
    with pytest.raises(NotImplementedError):
        Encoder(0, 48000, 2, 16, 2048, "s16le", bridge=_bridge(kind)).set_ecc(True, (96, 24))


def test_decoder_resync_and_truncation(kind):
    pcm = synth.to_pcm(synth.harmonic_mix(5000, 2, 48000, seed=1), "s16le").tobytes()
    p = dict(profile=4, srate=48000, channels=2, bits=16, frame_size=1024, pcm_format="s16le")
    stream, _ = _encode(kind, pcm, 4096, p)
    want = fo.decode_stream(stream)
    dec = Decoder(bridge=_bridge(kind))
    d = dec.process(b"garbage-before-the-first-frame" + stream[:-100])
    assert d.frames == 4 and not dec.is_empty()
    assert np.array_equal(d.pcm, want[:4 * 1024])
    assert dec.process(b"").frames == 0 and dec.broken_frame and dec.is_empty()


def test_decoder_one_undecodable_compact_frame_is_silence(kind):
    """profile1.py:59-60: a frame whose deflate body does not inflate decodes to zeros -- also when it is the only frame
    of a process() call (the device then gets a batch of empty bodies)."""
    arr = load_npz("g3_p1_streams.npz")
    g3 = load_json("g3_streams.json")
    c = [c for c in g3["cases"] if c["params"]["profile"] == 1][0]
    stream = bytearray(arr[f"{c['name']}_stream"].tobytes())
    f0, h0 = fo.asfh_parse(bytes(stream), 0)
    end0 = h0 + f0["frmbytes"]
    f1, h1 = fo.asfh_parse(bytes(stream), end0)
    for i in range(end0 + h1, end0 + h1 + f1["frmbytes"]):
        stream[i] = 0xFF                                     # frame 1: not a deflate stream any more
    end1 = end0 + h1 + f1["frmbytes"]
    dec = Decoder(bridge=_bridge(kind))
    a = dec.process(bytes(stream[:end0]))
    b = dec.process(bytes(stream[end0:end1]))                # the damaged frame, alone in its call
    assert a.frames == 1 and b.frames == 1
    ch = c["params"]["channels"]
    # what the reference does with the same bytes
    want = fo.decode_stream(bytes(stream[:end1]))
    got = np.concatenate([a.pcm.reshape(-1, ch), b.pcm.reshape(-1, ch), dec.flush().pcm.reshape(-1, ch)])
    assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-12


@pytest.mark.parametrize("profile,frame_size,tail", [(0, 8192, 5121), (0, 8192, 8191), (0, 16384, 6000), (0, 4096, 1),
                                                     (1, 16384, 9000), (1, 8192, 4500), (1, 16384, 16000)])
def test_flush_tails_of_large_frames(kind, profile, frame_size, tail):
    """A clip's last frame is whatever is left (encoder.py:72-93): with 8192 / 16384-sample frames the tail is an odd
    length (profile 0) or pads to a non-power-of-two compact size (profile 1: 9000 -> 10240, 4500 -> 5120,
    16000 -> 16384) that no LDS-resident kernel holds in stereo -- these used to raise FRAD_E_UNSUPPORTED in flush()."""
    C = 2
    pcm = synth.to_pcm(synth.harmonic_mix(frame_size + tail, C, 48000, seed=tail), "s16le").tobytes()
    p = dict(profile=profile, srate=48000, channels=C, bits=16 if profile else 32, frame_size=frame_size, pcm_format="s16le")
    out, samples = _encode(kind, pcm, 1 << 20, p)
    ref = fo.encode_stream(pcm, **p)
    assert samples == frame_size + tail
    a, b = fo.decode_stream(out), fo.decode_stream(ref)
    assert a.shape == b.shape and (profile == 1 or a.shape == (frame_size + tail, C))
    if profile == 0:
        assert len(out) == len(ref)
        assert np.max(np.abs(a - b)) <= 1e-9                  # stored float32 words equal up to rounding ties
    else:
        psnr = 10 * np.log10(1.0 / max(np.mean((a - b) ** 2), 1e-300))
        assert psnr > 100, psnr
    got, frames = _decode(kind, ref, 1 << 20, C)
    assert frames == 2 and got.shape == b.shape
    assert np.max(np.abs(got - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))


def test_encoder_is_transactional_on_device_errors():
    """A failing launch must not consume input: buffer and carry state are committed only on success."""
    class Boom(Exception):
        pass

    class FailingBridge:
        def __init__(self):
            from helpers import OracleBridge
            self.inner, self.fail = OracleBridge(), True

        def __getattr__(self, name):
            fn = getattr(self.inner, name)

            def call(*a, **k):
                if self.fail:
                    raise Boom(name)
                return fn(*a, **k)
            return call
    pcm = synth.to_pcm(synth.harmonic_mix(5000, 2, 48000, seed=2), "s16le").tobytes()
    br = FailingBridge()
    enc = Encoder(0, 48000, 2, 32, 1024, "s16le", bridge=br)
    with pytest.raises(Boom):
        enc.process(pcm)
    assert enc.buffer == b"" and not enc.have_carry
    br.fail = False
    out = enc.process(pcm).buf + enc.flush().buf
    want = fo.encode_stream(pcm, profile=0, srate=48000, channels=2, bits=32, frame_size=1024, pcm_format="s16le")
    assert out == want


@pytest.mark.parametrize("fmt", ["s16le", "f32le", "s32be"])
def test_decoder_output_format(kind, fmt):
    """Decoder(out_format=...) == from_f64(Decoder().pcm, fmt).astype(fmt) as the reference's caller computes it after
    every process() (src/decoder.py:23), for lossless and lossy streams incl. the tail frame and the flush."""
    from frad_python_amd.backend.pcmformat import ff_format_to_numpy_type
    dt = ff_format_to_numpy_type(fmt)
    pcm = synth.to_pcm(synth.harmonic_mix(7000, 2, 48000, seed=5), "s16le").tobytes()
    for p in (dict(profile=4, srate=48000, channels=2, bits=16, frame_size=1024, pcm_format="s16le"),
              dict(profile=0, srate=48000, channels=2, bits=32, frame_size=2048, pcm_format="s16le"),
              dict(profile=1, srate=48000, channels=2, bits=16, frame_size=2048, pcm_format="s16le", overlap_ratio=16)):
        stream = fo.encode_stream(pcm, **p)
        plain, frames = _decode(kind, stream, 5000, 2)
        dec = Decoder(bridge=_bridge(kind), out_format=fmt)
        got = []
        for i in range(0, len(stream), 5000):
            got.append(dec.process(stream[i:i + 5000]).pcm.reshape(-1, 2))
        got.append(dec.flush().pcm.reshape(-1, 2))
        assert all(g.dtype == dt for g in got if g.size)
        got = np.concatenate([g for g in got if g.size]).astype(dt)
        with np.errstate(all="ignore"):
            want = fo.from_f64(plain, dt).astype(dt)
        assert got.dtype == dt and got.shape == want.shape, (p["profile"], got.dtype, got.shape, want.shape)
        assert got.tobytes() == want.tobytes(), p["profile"]


def test_lossless_frame_length_comes_from_the_payload(kind):
    """profile0/4.digital ignore the header's fsize: they unpack what the payload holds (profile4.py:43-63).  A header
    that lies about fsize must decode like the reference."""
    pcm = synth.to_pcm(synth.harmonic_mix(3000, 2, 48000, seed=6), "s16le").tobytes()
    p = dict(profile=4, srate=48000, channels=2, bits=16, frame_size=1024, pcm_format="s16le")
    stream = bytearray(fo.encode_stream(pcm, **p))
    at = bytes(stream).find(b"\xff\xd0\xd2\x98")
    assert int.from_bytes(stream[at + 24:at + 28], "big") == 1024
    stream[at + 24:at + 28] = (777).to_bytes(4, "big")          # first frame's header now claims 777 sample-frames
    want = fo.decode_stream(bytes(stream))
    got, frames = _decode(kind, bytes(stream), 4096, 2)
    assert got.shape == want.shape == (3000, 2) and np.array_equal(got, want)


@pytest.mark.gpu
def test_streaming_api_under_random_chunking():
    """Encoder / Decoder on the HIP bridge fed in random chunk sizes (the table-driven header scan, strided batched
    decode, carry-over of half headers / half frames, tails and flushes): what comes out must not depend on the chunking
    and must match the oracle's one-shot encode / decode."""
    rng = np.random.default_rng(2026)
    for it in range(12):
        profile = int(rng.choice([0, 4, 1, 0]))
        C = int(rng.choice([1, 2, 2, 3]))
        fsize = int(rng.choice([256, 1024, 2048, 2048, 4096] if profile != 1 else [1024, 2048, 2048]))
        n = int(rng.integers(1, 9) * fsize + rng.integers(0, fsize))
        fmt = str(rng.choice(["s16le", "s16le", "f32le", "u8"])) if profile != 1 else "s16le"
        bits = int(rng.choice([16, 32, 64])) if profile != 1 else 16
        pcm = synth.to_pcm(synth.harmonic_mix(n, C, 48000, seed=it) * 0.7, fmt).tobytes()
        p = dict(profile=profile, srate=48000, channels=C, bits=bits, frame_size=fsize, pcm_format=fmt,
                 little_endian=bool(rng.integers(0, 2)), overlap_ratio=int(rng.choice([0, 8, 16])) if profile == 1 else 0)
        ref = fo.encode_stream(pcm, **p)
        want = fo.decode_stream(ref)
        # encode in random chunks
        enc = Encoder(profile, 48000, C, bits, fsize, fmt, bridge=_bridge("gpu"))
        enc.set_little_endian(p["little_endian"]); enc.set_overlap_ratio(p["overlap_ratio"]); enc.set_loss_level(0.5)
        out, pos, samples = b"", 0, 0
        while pos < len(pcm):
            step = int(rng.choice([1, 37, 4096, 65536, 1 << 20])) * (1 if rng.integers(0, 2) else 3)
            r = enc.process(pcm[pos:pos + step]); out += r.buf; samples += r.samples; pos += step
        r = enc.flush(); out += r.buf; samples += r.samples
        assert samples == n, (it, p)
        got_from_ours = fo.decode_stream(out)
        assert got_from_ours.shape == want.shape, (it, p)
        if profile == 4:
            assert out == ref, (it, p)
        elif profile == 0:
            tol = {16: 2e-3, 32: 1e-6, 64: 1e-12}[bits]
            assert np.max(np.abs(got_from_ours - want)) <= tol, (it, p)
        else:
            assert 10 * np.log10(1.0 / max(np.mean((got_from_ours - want) ** 2), 1e-300)) > 100, (it, p)
        # decode the reference stream in random chunks
        dec = Decoder(bridge=_bridge("gpu"))
        pieces, pos, frames = [], 0, 0
        while pos < len(ref):
            step = int(rng.choice([1, 5, 33, 1000, 50000, 1 << 20]))
            d = dec.process(ref[pos:pos + step]); pieces.append(d.pcm.reshape(-1, C)); frames += d.frames; pos += step
        pieces.append(dec.flush().pcm.reshape(-1, C))
        got = np.concatenate([x for x in pieces if x.size])
        assert got.shape == want.shape, (it, p, got.shape, want.shape)
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want))), (it, p)
