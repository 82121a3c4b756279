"""The depth-escalation branch of the product path (profile0.py:24-26, profile4.py:21-23) on the MI355X: frames whose
transform exceeds the storage float's range are re-dispatched at the deeper format the reference settles on, through
``core.analogue_batch(check_overflow=True)``, ``HipBridge`` and the streaming ``Encoder``.  Expected payloads and
depth indices are the reference's own (golden G5 ``esc*``); bigger batches are checked against the oracle."""
import numpy as np
import pytest

from frad_python_amd import Encoder, synth
from oracle import frad_oracle as fo
from helpers import word_mismatches

pytestmark = pytest.mark.gpu
CASES = (("esc16", 16), ("esc32", 32), ("esc12", 12), ("esc24", 24))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_g5_escalation_fixtures_through_analogue_batch(g5):
    from frad_python_amd import core
    for name, bits in CASES:
        x = np.ascontiguousarray(g5[f"{name}_in"], "<f8")
        n, c = x.shape
        for prof in (0, 4):
            enc = core.analogue_batch(prof, _dev(x.view(np.uint8).reshape(-1)), "f64le", 1, n, c, bits, False, check_overflow=True)
            got, used = enc.frame_bytes(0)
            idx = int(g5[f"{name}_p{prof}_idx"])
            assert used == fo.DEPTHS[idx] and (0 in enc.escalated) == (used != bits), (name, prof, used)   # (esc32 only overflows as raw PCM: its DCT bins fit float32)
            want = g5[f"{name}_p{prof}_frad"].tobytes()
            if prof == 4:
                assert got == want, (name, prof)
            else:                                             # value contract of profile 0: identical words up to a rounding tie
                assert len(got) == len(want) and word_mismatches(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8), used) <= 1, (name, prof)


def test_g5_escalation_fixtures_through_the_encoder(g5):
    from frad_python_amd.bridge import HipBridge
    for name, bits in CASES:
        x = np.ascontiguousarray(g5[f"{name}_in"])
        n, c = x.shape
        for prof in (0, 4):
            p = dict(profile=prof, srate=48000, channels=c, bits=bits, frame_size=n, pcm_format="f64be")
            pcm = np.concatenate([x, x * 1e-9, x]).astype(">f8").tobytes()       # escalating, quiet, escalating frame
            enc = Encoder(prof, 48000, c, bits, n, "f64be", bridge=HipBridge())
            out = enc.process(pcm).buf + enc.flush().buf
            ref = fo.encode_stream(pcm, **p)
            assert len(out) == len(ref), (name, prof)
            if prof == 4:
                assert out == ref, (name, prof)
            else:
                a, b = np.frombuffer(out, np.uint8), np.frombuffer(ref, np.uint8)
                assert np.count_nonzero(a != b) <= 12, (name, prof)              # at most one tie per frame (+ its CRC bytes)
                assert np.max(np.abs(fo.decode_stream(out) - fo.decode_stream(ref))) <= 1e-9 * 1e6, (name, prof)
            # header depth indices: frames 0 and 2 escalated, frame 1 did not
            depth, pos = [], 0
            while pos < len(out):
                f, hlen = fo.asfh_parse(out, pos)
                depth.append(f["depth_idx"]); pos += hlen + f["frmbytes"]
            assert depth[0] == depth[2] == int(g5[f"{name}_p{prof}_idx"]) and depth[1] == fo.DEPTHS.index(bits), (name, prof, depth)


@pytest.mark.parametrize("fmt,N,C", [("f64le", 2048, 2), ("f64le", 512, 1), ("s32be", 2048, 2)])
def test_one_escalating_frame_among_many(fmt, N, C):
    """A batch of 33 frames in which a few overflow float16 storage (the whole-stream device path must fall through to
    the per-frame framer for them, bridge.lossless_encode_stream -> None)."""
    from frad_python_amd import core
    from frad_python_amd.bridge import HipBridge
    rng = np.random.default_rng(99)
    F, bits = 33, 16
    x = rng.uniform(-1, 1, (F * N, C))
    if fmt == "s32be":
        raw = synth.to_pcm(x, fmt)                            # unscaled big-endian ints (the reference's quirk): EVERY frame escalates
        hot = list(range(F))
    else:
        hot = [5, 17]
        x[17 * N + 3, 0] = 1.0e6                              # -> 24 bit
        x[5 * N + 100, C - 1] = 3.0e38                        # DCT bins stay below float32's range -> 24 bit as well
        raw = synth.to_pcm(x, fmt)
    dt = fo.pcm_dtype(fmt)
    for prof in (0, 4):
        enc = core.analogue_batch(prof, _dev(raw.view(np.uint8).reshape(-1)), fmt, F, N, C, bits, False, check_overflow=True)
        mism = words = 0
        want_hot = []
        for f in range(F):
            frame = fo.to_f64(raw[f * N:(f + 1) * N], dt)
            frad, idx, ch, sr = (fo.p0_analogue if prof == 0 else fo.p4_analogue)(frame, bits, 48000, False)
            if fo.DEPTHS[idx] != bits:
                want_hot.append(f)                            # (a lone 1e6 sample overflows float16 as PCM, not as 2048 DCT bins)
            got, used = enc.frame_bytes(f)
            assert used == fo.DEPTHS[idx], (prof, f, used, idx)
            if prof == 4:
                assert got == frad, (prof, f)
            else:
                assert len(got) == len(frad)
                mism += word_mismatches(np.frombuffer(got, np.uint8), np.frombuffer(frad, np.uint8), used); words += N * C
        assert mism <= max(2, int(1e-5 * words)), (prof, mism, words)
        assert sorted(enc.escalated) == want_hot and want_hot and set(want_hot) <= set(hot), (prof, sorted(enc.escalated), want_hot)
        # the same through the streaming encoder
        p = dict(profile=prof, srate=48000, channels=C, bits=bits, frame_size=N, pcm_format=fmt)
        e = Encoder(prof, 48000, C, bits, N, fmt, bridge=HipBridge())
        out = e.process(raw.tobytes()).buf + e.flush().buf
        ref = fo.encode_stream(raw.tobytes(), **p)
        assert len(out) == len(ref), prof
        if prof == 4:
            assert out == ref
        else:
            d1, d2 = fo.decode_stream(out), fo.decode_stream(ref)
            assert np.max(np.abs(d1 - d2)) <= 1e-9 * max(1.0, np.max(np.abs(d2))), prof


def test_checked_analogue_sets_the_sticky_overflow_flag():
    """frad_p0_analogue_checked == frad_p0_analogue + frad_p0_overflow_scan (profile0.py:24-26 over a batch), for the
    wave kernels (test fused into the transform) and for the other geometries (scan launched behind it)."""
    import torch
    from frad_python_amd import core
    rng = np.random.default_rng(3)
    for (N, C, fmt) in ((2048, 2, "f64le"), (2048, 1, "f64le"), (1024, 2, "f64le"), (896, 2, "f64le")):
        F = 9
        x = rng.uniform(-1, 1, (F * N, C))
        for hot in (False, True):
            if hot:
                x[4 * N + 7, 0] = 3.0e38                      # frame 4 overflows float16 storage
            raw = _dev(np.ascontiguousarray(x, "<f8").view(np.uint8).reshape(-1))
            flag = torch.zeros((), dtype=torch.int32, device="cuda:0")
            a = core.analogue_batch(0, raw, fmt, F, N, C, 16, False, check_overflow=False, overflow_flag=flag)
            b = core.analogue_batch(0, raw, fmt, F, N, C, 16, False, check_overflow=False)
            assert torch.equal(a.payload, b.payload) and torch.equal(a.absmax, b.absmax)
            assert int(flag.item()) == int(hot), (N, C, hot)
            assert bool((a.absmax > 65504).any()) == hot
        flag = torch.ones((), dtype=torch.int32, device="cuda:0")           # sticky: a clean batch leaves it set
        core.analogue_batch(0, _dev(np.zeros(F * N * C, "<f8").view(np.uint8)), fmt, F, N, C, 16, False, check_overflow=False, overflow_flag=flag)
        assert int(flag.item()) == 1
