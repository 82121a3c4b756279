"""CPU restatement of the FrAD transform core -- TEST INFRASTRUCTURE ONLY.

This module is the *oracle* for the MI355X-native FrAD transform core.  It restates, in
NumPy/SciPy, the reference algorithm of H4n-uL/FrAD_Python for the one hot path this
repository accelerates.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
``frad_python_amd`` never does (a product path through the oracle voids every parity claim).

Parity pin: every function here is checked bit-for-bit (profile 4, all pack/unpack, profile 0
and profile 1 on the generating host) against outputs of the reference itself, produced in
the build container by ``oracle/gen_golden.py`` and committed under ``tests/golden/``.
The Fourier arithmetic itself lives in a third-party dependency of the reference
(``scipy.fft.dct/idct`` = pocketfft; the reference does not pin a version, the image ships
scipy 1.15.3 / numpy 2.2.6) -- the oracle calls the very same library function, so the
oracle == reference identity holds on any host with that scipy.

All paths cited as ``ref:`` are relative to ``/root/reference/src/libfrad/``.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np
from scipy.fft import dct as _dct, idct as _idct

# ---------------------------------------------------------------------------------------
# constants (ref: fourier/profile0.py:4-13, fourier/profile1.py:7, fourier/profiles.py:1-34,
#            fourier/tools/p1tools.py:4-13, common.py:1-2)
# ---------------------------------------------------------------------------------------
DEPTHS = (12, 16, 24, 32, 48, 64)                 # profiles 0 and 4
P1_DEPTHS = (8, 12, 16, 24, 32, 48, 64)           # profile 1 (only sets the 2^(bits-1) scale)
_STORE = {64: "f8", 48: "f8", 32: "f4", 24: "f4", 16: "f2", 12: "f2"}
_ESCALATE = {12: 16, 16: 24, 24: 32, 32: 48, 48: 64, 64: 128}
FLOAT_MAX = {b: float(np.finfo(_STORE[b]).max) for b in DEPTHS}

LOSSLESS = (0, 4)
COMPACT = (1, 2)
COMPACT_SRATES = (96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000)
COMPACT_SAMPLES = tuple(m << s for s in range(8) for m in (128, 160, 192, 224))

BAND_EDGES_HZ = (0, 200, 400, 600, 800, 1000, 1200, 1400, 1600, 2000, 2400, 2800, 3200, 4000,
                 4800, 5600, 6800, 8000, 9600, 12000, 15600, 20000, 24000, 28800, 34400, 40800,
                 48000, (1 << 32) - 1)
N_BANDS = len(BAND_EDGES_HZ) - 1                   # 27
SPREAD_ALPHA = 0.8
QUANT_ALPHA = 0.75

FRM_SIGN = b"\xff\xd0\xd2\x98"

PCM_FORMATS = {
    "u8": "u1", "u16be": ">u2", "u16le": "<u2", "u32be": ">u4", "u32le": "<u4", "u64be": ">u8",
    "u64le": "<u8", "s8": "i1", "s16be": ">i2", "s16le": "<i2", "s32be": ">i4", "s32le": "<i4",
    "s64be": ">i8", "s64le": "<i8", "f16be": ">f2", "f16le": "<f2", "f32be": ">f4",
    "f32le": "<f4", "f64be": ">f8", "f64le": "<f8",
}


def compact_valid_srate(srate: int) -> int:
    """ref: fourier/profiles.py:7-8"""
    return min(s for s in COMPACT_SRATES if s >= srate)


def compact_samples_min_ge(n: int) -> int:
    """ref: fourier/profiles.py:26-27"""
    return min(s for s in COMPACT_SAMPLES if s >= n)


# ---------------------------------------------------------------------------------------
# R1  PCM <-> float   (ref: backend/pcmformat.py:4-62)
# ---------------------------------------------------------------------------------------
def pcm_dtype(fmt: str) -> np.dtype:
    """ref: backend/pcmformat.py:4-32 (unknown names are an error there too)."""
    try:
        return np.dtype(PCM_FORMATS[fmt.lower()])
    except KeyError:
        raise ValueError(f"Invalid format: {fmt}")


def to_f64(pcm: np.ndarray, dt: np.dtype, *, be_int_quirk: bool = True) -> np.ndarray:
    """ref: backend/pcmformat.py:34-47.

    Floats are returned *unchanged* (no widening, :35).  Native-order integers are divided by
    2^(w-1) (unsigned: then minus one).  The reference compares ``dtype == np.int16`` etc.,
    which is False for a byte-swapped dtype, so on a little-endian host big-endian integer
    PCM falls through every branch and is returned unscaled (:37-45); ``be_int_quirk=True``
    reproduces that, ``False`` gives the arithmetic the author evidently meant.
    """
    dt = np.dtype(dt)
    if dt.kind == "f":
        return pcm
    if be_int_quirk and not dt.isnative:
        return pcm
    half = float(1 << (8 * dt.itemsize - 1))
    out = pcm.astype(np.float64) / half
    if dt.kind == "u":
        out = out - 1
    return out


def from_f64(pcm: np.ndarray, dt: np.dtype) -> np.ndarray:
    """ref: backend/pcmformat.py:49-62 (truncating ``astype``, native-order ints only)."""
    dt = np.dtype(dt)
    if dt.kind == "f" or not dt.isnative:
        return pcm
    half = float(1 << (8 * dt.itemsize - 1))
    if dt.kind == "u":
        return ((pcm + 1) * half).astype(dt)
    return (pcm * half).astype(dt)


# ---------------------------------------------------------------------------------------
# bit-depth pack / unpack  (ref: fourier/profile0.py:28-42, 51-63 == profile4.py:25-39, 48-60)
# ---------------------------------------------------------------------------------------
def payload_bytes(n_values: int, bits: int) -> int:
    return (n_values * 3 + 1) // 2 if bits == 12 else n_values * bits // 8


def pack_floats(flat: np.ndarray, bits: int, little_endian: bool) -> bytes:
    """RN cast to the storage float, then byte/nibble truncation.

    ref: profile0.py:29-30 (cast; 12-bit is always big-endian), :35-36 (24/48: keep the
    high-order 3 of 4 / 6 of 8 bytes on either endianness), :37-41 (12: top three hex
    nibbles of each f16, odd nibble count padded with one zero nibble).
    """
    be = (not little_endian) or bits % 8 != 0
    raw = np.ascontiguousarray(flat).astype((">" if be else "<") + _STORE[bits])
    if bits in (16, 32, 64):
        return raw.tobytes()
    if bits in (24, 48):
        w = raw.dtype.itemsize
        keep = bits // 8
        b = raw.view(np.uint8).reshape(-1, w)
        return (b[:, :keep] if be else b[:, w - keep:]).tobytes()
    if bits == 12:
        h = (raw.view(">u2").astype(np.uint16) >> 4).astype(np.uint32)     # 12-bit codes
        n = h.size
        if n & 1:
            h = np.concatenate([h, np.zeros(1, np.uint32)])
        a, b = h[0::2], h[1::2]
        out = np.empty((a.size, 3), np.uint8)
        out[:, 0] = a >> 4
        out[:, 1] = ((a & 0xF) << 4) | (b >> 8)
        out[:, 2] = b & 0xFF
        return out.tobytes()[: payload_bytes(n, 12)]
    raise ValueError("Illegal bits value.")


def unpack_floats(frad: bytes, bits: int, little_endian: bool) -> np.ndarray:
    """Inverse of :func:`pack_floats`, widened to float64 (ref: profile0.py:51-63)."""
    be = (not little_endian) or bits % 8 != 0
    e = ">" if be else "<"
    buf = np.frombuffer(frad, np.uint8)
    if bits in (16, 32, 64):
        n = buf.size // (bits // 8)
        return np.frombuffer(frad, e + _STORE[bits], n).astype(np.float64)
    if bits in (24, 48):
        keep, w = bits // 8, bits // 6
        # the reference slices every `keep` bytes; a ragged tail becomes a short (invalid) word
        n = buf.size // keep
        full = np.zeros((n, w), np.uint8)
        if be:
            full[:, :keep] = buf[: n * keep].reshape(n, keep)
        else:
            full[:, w - keep:] = buf[: n * keep].reshape(n, keep)
        return full.reshape(-1).view(e + _STORE[bits]).astype(np.float64)
    if bits == 12:
        nib = buf.size * 2
        if nib % 3:
            nib -= 1                       # strip the pad nibble (ref: profile0.py:57)
        n = nib // 3
        b = np.concatenate([buf, np.zeros(3, np.uint8)]).astype(np.uint32)
        i = np.arange(n)
        p = (i * 3) >> 1                   # byte holding the first nibble
        even = (i & 1) == 0
        code = np.where(even, (b[p] << 4) | (b[p + 1] >> 4), ((b[p] & 0xF) << 8) | b[p + 1])
        return (code << 4).astype(">u2").view(">f2").astype(np.float64)
    raise ValueError("Illegal bits value.")


def _escalate(absmax: float, bits: int) -> int:
    """ref: profile0.py:24-26.  NaN compares False, so NaN never escalates; +Inf runs off the
    end of the table (the reference then dies in ``DEPTHS.index(128)``)."""
    absmax = float(absmax)
    while absmax > FLOAT_MAX[bits]:
        bits = _ESCALATE[bits]
        if bits == 128:
            raise OverflowError("Overflow with reaching the max bit depth.")
    return bits


def _scrub(x: np.ndarray) -> np.ndarray:
    """ref: profile0.py:66 / profile4.py:63 -- NaN and +-Inf become 0."""
    return np.where(np.isfinite(x), x, 0.0)


# ---------------------------------------------------------------------------------------
# R9  the Fourier arithmetic (third-party in the reference: scipy.fft, pocketfft)
# ---------------------------------------------------------------------------------------
def dct_channels(pcm: np.ndarray) -> np.ndarray:
    """[n, C] -> [C, n], DCT-II with norm='forward' per channel (ref: profile0.py:21).

    float32/float16 input stays in float32 arithmetic inside pocketfft, exactly as in the
    reference (its ``to_f64`` does not widen floats).  One batched call over contiguous rows
    is bitwise identical to the reference's per-channel loop (checked in the golden tests).
    """
    return _dct(np.ascontiguousarray(pcm.T), axis=1, norm="forward")


def idct_channels(freqs: np.ndarray) -> np.ndarray:
    """[C, n] -> [n, C] float64 C-contiguous (ref: profile0.py:69)."""
    return np.ascontiguousarray(_idct(np.ascontiguousarray(freqs), axis=1, norm="forward").T)


# ---------------------------------------------------------------------------------------
# R3/R4  profile 0, R5 profile 4  (frame-in -> frame-out, same signatures as the reference)
# ---------------------------------------------------------------------------------------
def p0_analogue(pcm: np.ndarray, bits: int, srate: int, little_endian: bool):
    """ref: fourier/profile0.py:14-44."""
    if bits not in DEPTHS:
        bits = 16
    channels = pcm.shape[1]
    freqs = dct_channels(pcm)
    bits = _escalate(np.max(np.abs(freqs)), bits)
    frad = pack_floats(freqs.T.ravel(), bits, little_endian)
    return frad, DEPTHS.index(bits), channels, srate


def p0_digital(frad: bytes, fb: int, channels: int, little_endian: bool) -> np.ndarray:
    """ref: fourier/profile0.py:46-69."""
    freqs = unpack_floats(frad, DEPTHS[fb], little_endian).reshape(-1, channels).T
    return idct_channels(_scrub(freqs))


def p0_analogue_batch(pcm: np.ndarray, n_frames: int, N: int, C: int, bits: int, *, little_endian: bool = False,
                      fmt: str = "s16le", workers: int = 1) -> np.ndarray:
    """All-cores CPU baseline of bench.py: :func:`p0_analogue` over ``n_frames`` consecutive frames of
    ``pcm`` [n_frames*N, C] in ONE scipy call per stage -- the transform runs over the channel rows of every frame at once
    (``workers`` threads; bitwise the per-frame result: same per-row pocketfft plan) and the cast / pack is vectorised.
    Whole-byte depths without escalation only (the bench signal never overflows).  -> uint8 [n_frames, N*C*bits/8]."""
    if bits not in (16, 32, 64):
        raise ValueError("batched baseline: 16 / 32 / 64-bit storage")
    x = to_f64(pcm.reshape(n_frames, N, C), pcm_dtype(fmt))
    rows = np.ascontiguousarray(x.transpose(0, 2, 1))                   # [F, C, N]: contiguous rows, as dct_channels feeds them
    freqs = _dct(rows, axis=2, norm="forward", workers=workers)
    if np.max(np.abs(freqs)) > FLOAT_MAX[bits]:
        raise OverflowError("a frame needs a deeper format: use p0_analogue")
    flat = np.ascontiguousarray(freqs.transpose(0, 2, 1)).reshape(n_frames, N * C)      # bin-major / channel-minor
    e = "<" if little_endian else ">"
    return flat.astype(e + _STORE[bits]).view(np.uint8).reshape(n_frames, -1)


def p0_digital_batch(payload: np.ndarray, n_frames: int, N: int, C: int, bits: int, *, little_endian: bool = False,
                     workers: int = 1) -> np.ndarray:
    """:func:`p0_digital` over a batch of equal frames (see :func:`p0_analogue_batch`).  -> float64 [n_frames, N, C]."""
    if bits not in (16, 32, 64):
        raise ValueError("batched baseline: 16 / 32 / 64-bit storage")
    e = "<" if little_endian else ">"
    vals = np.ascontiguousarray(payload).view(e + _STORE[bits]).astype(np.float64).reshape(n_frames, N, C)
    rows = _scrub(np.ascontiguousarray(vals.transpose(0, 2, 1)))
    return np.ascontiguousarray(_idct(rows, axis=2, norm="forward", workers=workers).transpose(0, 2, 1))


def p4_analogue(pcm: np.ndarray, bits: int, srate: int, little_endian: bool):
    """ref: fourier/profile4.py:14-41."""
    if bits not in DEPTHS:
        bits = 16
    channels = pcm.shape[1]
    bits = _escalate(np.max(np.abs(pcm)), bits)
    return pack_floats(pcm.ravel(), bits, little_endian), DEPTHS.index(bits), channels, srate


def p4_digital(frad: bytes, fb: int, channels: int, little_endian: bool) -> np.ndarray:
    """ref: fourier/profile4.py:43-63."""
    return _scrub(unpack_floats(frad, DEPTHS[fb], little_endian).reshape(-1, channels))


# ---------------------------------------------------------------------------------------
# R6/R7  profile 1 (psychoacoustic quantiser), pre-entropy part and the host entropy stage
# ---------------------------------------------------------------------------------------
def band_edges(dlen: int, srate: int) -> list[int]:
    """Bin index of every band edge: Python ``round`` (half-to-even) of dlen/(srate/2)*Hz
    (ref: p1tools.py:15-16).  Not clipped to dlen -- callers clip (ref: p1tools.py:38-39)."""
    return [round(dlen / (srate / 2) * hz) for hz in BAND_EDGES_HZ]


def hearing_threshold(band: int) -> float:
    """ATH at the band centre (ref: p1tools.py:25-28)."""
    f = (BAND_EDGES_HZ[band] + BAND_EDGES_HZ[band + 1]) / 2
    khz = f / 1000.0
    return 10.0 ** ((3.64 * khz ** -0.8 - 6.5 * np.exp(-0.6 * (khz - 3.3) ** 2.0) + 1e-3 * (khz ** 4.0)) / 20)


def mask_thresholds(freqs: np.ndarray, srate: int, loss_level: float, alpha: float = SPREAD_ALPHA) -> np.ndarray:
    """27 per-band masking thresholds of one channel (ref: p1tools.py:18-33).

    The loop stops at the first band whose bin slice is empty and leaves the rest zero."""
    mag = np.abs(freqs)
    edges = band_edges(len(mag), srate)
    thres = np.zeros(N_BANDS)
    for i in range(N_BANDS):
        sub = mag[edges[i]:edges[i + 1]]
        if len(sub) == 0:
            break
        rms_a = np.sqrt(np.mean(sub ** 2)) ** alpha
        thres[i] = max(rms_a, min(hearing_threshold(i), 1.0)) * loss_level
    return thres


def spread_thresholds(band_thres: np.ndarray, dlen: int, srate: int) -> np.ndarray:
    """Per-bin divisor: linear ramps between consecutive band *starts* over bands 0..25,
    endpoint excluded; bins past the last start stay 0 (ref: p1tools.py:35-41)."""
    edges = [min(e, dlen) for e in band_edges(dlen, srate)]
    out = np.zeros(dlen)
    for i in range(N_BANDS - 1):
        a, b = edges[i], edges[i + 1]
        out[a:b] = np.linspace(band_thres[i], band_thres[i + 1], b - a, endpoint=False)
    return out


def quant(x):
    """ref: p1tools.py:43"""
    return np.sign(x) * np.abs(x) ** QUANT_ALPHA


def dequant(x):
    """ref: p1tools.py:44"""
    return np.sign(x) * np.abs(x) ** (1 / QUANT_ALPHA)


def p1_analogue_pre(pcm: np.ndarray, bits: int, srate: int, loss_level: float):
    """Profile 1 up to the GPU/host split point (ref: fourier/profile1.py:15-40).

    Returns ``(q, tq, aux)``: ``q`` int64 [dlen*C] bin-major/channel-minor, ``tq`` int64
    [27*C] band-major/channel-minor and a dict of float intermediates for tolerance tests."""
    if bits not in P1_DEPTHS:
        bits = 16
    scale = 2.0 ** (bits - 1)
    dlen = compact_samples_min_ge(len(pcm))
    pcm = np.pad(pcm, ((0, dlen - len(pcm)), (0, 0)), mode="constant")
    srate = compact_valid_srate(srate)
    loss_level = max(abs(loss_level), 0.125)
    channels = pcm.shape[1]
    freqs = dct_channels(pcm)
    masked, thres = [], []
    for c in range(channels):
        t = mask_thresholds(freqs[c] * scale, srate, loss_level)
        div = spread_thresholds(t, dlen, srate)
        div = np.where(div == 0, np.inf, div)
        masked.append(freqs[c] / div)
        thres.append(t)
    masked, thres = np.array(masked), np.array(thres)
    q = quant(masked * scale).round().astype(int).T.ravel()
    tq = dequant(np.log(thres.clip(min=1.0)) / np.log(np.e / 2)).round().astype(int).T.ravel()
    return q, tq, {"freqs": freqs, "thres": thres, "bits": bits, "srate": srate, "dlen": dlen}


def golomb_encode(data: np.ndarray) -> bytes:
    """Exp-Golomb-Rice code of a signed int vector (ref: p1tools.py:49-60), vectorised.

    1 byte k = ceil(log2(max|v|)), then per value the zig-zag code z (v>0: 2v-1, else -2v)
    written as m zeros followed by the (m+k+1)-bit binary of z + 2^k, zero-padded to a byte."""
    data = np.asarray(data).astype(np.int64)
    if not data.size:
        return b"\x00"
    dmax = int(np.abs(data).max())
    k = int(np.ceil(np.log2(dmax))) if dmax else 0
    z = np.where(data > 0, 2 * data - 1, -2 * data).astype(np.uint64) + np.uint64(1 << k)
    nbits = np.floor(np.log2(z.astype(np.float64))).astype(np.int64) + 1        # z < 2^53 here
    # guard the float log2 at exact powers of two
    nbits += (z >> nbits.astype(np.uint64)) > 0
    nbits -= (z >> (nbits - 1).astype(np.uint64)) == 0
    m = nbits - (k + 1)
    total = m + nbits
    ends = np.cumsum(total)
    starts = ends - total
    bits_out = np.zeros(int(ends[-1]), np.uint8)
    # set the '1' bits of each binary code
    maxb = int(nbits.max())
    for j in range(maxb):
        sel = nbits > j
        bit = ((z[sel] >> (nbits[sel] - 1 - j).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)
        bits_out[(starts[sel] + m[sel] + j)] = bit
    return struct.pack("B", k) + np.packbits(bits_out).tobytes()


def golomb_decode(dbytes: bytes) -> np.ndarray:
    """ref: p1tools.py:62-74 (a trailing run of zero bits ends the stream; a code cut short
    by the end of the buffer is still parsed from the bits that are there)."""
    k = dbytes[0]
    data = "".join(f"{b:08b}" for b in dbytes[1:])
    pos, out = 0, []
    while pos < len(data):
        one = data.find("1", pos)
        if one < 0:
            break
        ln = 2 * (one - pos) + k + 1
        v = int(data[pos:pos + ln], 2) - (1 << k)
        out.append((v + 1) >> 1 if v & 1 else -(v >> 1))
        pos += ln
    return np.array(out)


def p1_pack(q: np.ndarray, tq: np.ndarray) -> bytes:
    """Host entropy stage (ref: fourier/profile1.py:43-50): '>I' len + Golomb(tq) + Golomb(q),
    raw deflate (wbits=-15) at zlib's default level."""
    tg, fg = golomb_encode(tq), golomb_encode(q)
    raw = struct.pack(">I", len(tg)) + tg + fg
    co = zlib.compressobj(zlib.Z_DEFAULT_COMPRESSION, zlib.DEFLATED, -15)
    return co.compress(raw) + co.flush()


def p1_unpack(frad: bytes):
    """ref: fourier/profile1.py:59-66 up to the Golomb decode; None on a corrupt deflate."""
    try:
        raw = zlib.decompress(frad, wbits=-15)
    except Exception:
        return None
    tlen = struct.unpack(">I", raw[:4])[0]
    return golomb_decode(raw[4 + tlen:]), golomb_decode(raw[4:4 + tlen])


def p1_analogue(pcm, bits, srate, loss_level):
    """ref: fourier/profile1.py:15-52."""
    q, tq, aux = p1_analogue_pre(pcm, bits, srate, loss_level)
    return p1_pack(q, tq), P1_DEPTHS.index(aux["bits"]), pcm.shape[1], aux["srate"]


def p1_digital_post(q: np.ndarray, tq: np.ndarray, fb: int, channels: int, srate: int, fsize: int) -> np.ndarray:
    """Profile 1 from the decoded int arrays on (ref: fourier/profile1.py:65-77)."""
    scale = 2.0 ** (P1_DEPTHS[fb] - 1)
    f = dequant(np.asarray(q).astype(float)) / scale
    t = np.power(np.e / 2, quant(np.asarray(tq).astype(float)))
    f = np.pad(f, (0, max(0, fsize * channels - len(f))), "constant")
    t = np.pad(t, (0, max(0, fsize * channels - len(t))), "constant")
    t = t.reshape(-1, channels).T
    f = f.reshape(-1, channels).T
    freqs = np.array([f[c] * spread_thresholds(t[c], fsize, srate) for c in range(channels)])
    return idct_channels(freqs)


def p1_digital(frad: bytes, fb: int, channels: int, srate: int, fsize: int) -> np.ndarray:
    """ref: fourier/profile1.py:54-77."""
    u = p1_unpack(frad)
    if u is None:
        return np.zeros((fsize, channels))
    return p1_digital_post(u[0], u[1], fb, channels, srate, fsize)


# ---------------------------------------------------------------------------------------
# R8  decoder overlap-add (ref: decoder.py:28-46, backend/__init__.py:3)
# ---------------------------------------------------------------------------------------
def hanning_in_overlap(olap_len: int) -> np.ndarray:
    return 0.5 * (1 - np.cos(np.pi * np.arange(1, olap_len + 1) / (olap_len + 1)))


class OverlapAdd:
    """Stateful cross-fade of consecutive compact frames (ref: decoder.py:28-46, 110-114)."""

    def __init__(self):
        self.fragment = np.zeros((0, 0))
        self.prog = 0

    def push(self, frame: np.ndarray, compact: bool, ratio: int) -> np.ndarray:
        L = len(self.fragment)
        if L:
            w = hanning_in_overlap(L)
            n = min(L - self.prog, len(frame))
            i = np.arange(n) + self.prog
            frame[:n] = frame[:n] * w[i, None] + self.fragment[i] * w[L - 1 - i, None]
            self.prog += n
        if L <= self.prog:
            self.fragment, self.prog = np.zeros((0, 0)), 0
            if compact and ratio != 0:
                cut = len(frame) * (ratio - 1) // ratio
                self.fragment, frame = frame[cut:], frame[:cut]
        return frame

    def flush(self) -> np.ndarray:
        out, self.fragment, self.prog = self.fragment, np.zeros((0, 0)), 0
        return out


# ---------------------------------------------------------------------------------------
# host framing: ASFH (ref: tools/asfh.py:6-24, 51-96) -- used for stream-level parity
# ---------------------------------------------------------------------------------------
def asfh_write(frad: bytes, *, profile: int, ecc: bool, little_endian: bool, depth_idx: int,
               channels: int, srate: int, fsize: int, overlap_ratio: int = 0,
               ecc_ratio=(0, 0)) -> bytes:
    """One framed payload (ref: tools/asfh.py:51-73).  ECC-protected compact frames carry a
    CRC-16 the transform core never needs; they are outside this oracle."""
    pfb = (profile << 5) | (int(ecc) << 4) | (int(little_endian) << 3) | depth_idx
    head = FRM_SIGN + struct.pack(">I", len(frad)) + bytes([pfb])
    if profile in COMPACT:
        if ecc:
            raise NotImplementedError("compact + ECC framing is host-side and out of scope")
        css = ((channels - 1) << 10) | (COMPACT_SRATES.index(compact_valid_srate(srate)) << 6) \
            | (COMPACT_SAMPLES.index(compact_samples_min_ge(fsize)) << 1)
        head += struct.pack(">H", css) + bytes([max(overlap_ratio - 1, 0)])
    else:
        head += bytes([channels - 1, ecc_ratio[0], ecc_ratio[1]]) + struct.pack(">I", srate)
        head += b"\x00" * 8 + struct.pack(">I", fsize) + struct.pack(">I", zlib.crc32(frad))
    return head + frad


def asfh_force_flush(*, profile: int, ecc: bool, little_endian: bool, depth_idx: int,
                     channels: int, srate: int, fsize: int) -> bytes:
    """ref: tools/asfh.py:75-88 (lossless profiles emit nothing)."""
    if profile not in COMPACT:
        return b""
    pfb = (profile << 5) | (int(ecc) << 4) | (int(little_endian) << 3) | depth_idx
    css = ((max(channels, 1) - 1) << 10) | (COMPACT_SRATES.index(compact_valid_srate(srate)) << 6) \
        | (COMPACT_SAMPLES.index(compact_samples_min_ge(fsize)) << 1) | 1
    return FRM_SIGN + b"\x00" * 4 + bytes([pfb]) + struct.pack(">H", css) + b"\x00"


def encode_stream(pcm_bytes: bytes, *, profile: int, srate: int, channels: int, bits: int,
                  frame_size: int, pcm_format: str, little_endian: bool = False,
                  overlap_ratio: int = 0, loss_level: float = 0.5,
                  be_int_quirk: bool = True) -> bytes:
    """Whole-stream restatement of ``Encoder.process()`` (any chunking) followed by
    ``flush()``, ECC off (ref: encoder.py:35-51 overlap carry, :72-93 frame cut, :96-105
    dispatch + framing, :109-112).  Byte-for-byte the reference stream (golden G3).

    Frame cut: each frame takes ``N - len(carry)`` new sample-frames (compact: N rounded up
    to the table); ``process`` stops when fewer are buffered; ``flush`` then encodes whatever
    is left (carry + remainder, no new carry), appends a force-flush header after every
    frame it wrote and one more when nothing is left (lossless: those are empty)."""
    dt = pcm_dtype(pcm_format)
    step = dt.itemsize * channels
    pcm = np.frombuffer(pcm_bytes, dt, len(pcm_bytes) // step * channels).reshape(-1, channels)
    pcm = to_f64(pcm, dt, be_int_quirk=be_int_quirk)
    compact = profile in COMPACT
    if overlap_ratio != 0:
        overlap_ratio = max(2, min(256, overlap_ratio))
    loss_level = max(abs(loss_level), 0.125)
    n_eff = compact_samples_min_ge(frame_size) if compact else frame_size
    meta = dict(profile=profile, ecc=False, little_endian=little_endian)
    last = dict(depth_idx=0, channels=0, srate=0, fsize=0)
    out, pos, carry = [], 0, pcm[:0]

    def emit(frame):
        nonlocal last
        if profile == 1:
            frad, di, ch, sr = p1_analogue(frame, bits, srate, loss_level)
        elif profile == 4:
            frad, di, ch, sr = p4_analogue(frame, bits, srate, little_endian)
        else:
            frad, di, ch, sr = p0_analogue(frame, bits, srate, little_endian)
        last = dict(depth_idx=di, channels=ch, srate=sr, fsize=len(frame))
        out.append(asfh_write(frad, **meta, **last, overlap_ratio=overlap_ratio))

    while True:                                   # process(): whole frames only
        take = min(len(carry), n_eff)
        want = n_eff - take
        if len(pcm) - pos < want:
            break
        frame = np.concatenate([carry[:take], pcm[pos:pos + want]])
        pos += want
        carry = frame[len(frame) * (overlap_ratio - 1) // overlap_ratio:] \
            if compact and overlap_ratio > 1 else pcm[:0]
        emit(frame)
    while True:                                   # flush(): the remainder, then terminate
        take = min(len(carry), n_eff)
        new = pcm[pos:pos + n_eff - take]
        pos += len(new)
        frame = np.concatenate([carry[:take], new])
        carry = pcm[:0]
        if len(frame) == 0:
            out.append(asfh_force_flush(**meta, **last))
            break
        emit(frame)
        out.append(asfh_force_flush(**meta, **last))
    return b"".join(out)


def asfh_parse(buf: bytes, pos: int):
    """Parse one frame header at ``pos`` (must start with FRM_SIGN).  ref: tools/asfh.py:98-134.
    Returns ``(fields, header_len)``; ``fields['force_flush']`` marks a compact flush header."""
    assert buf[pos:pos + 4] == FRM_SIGN
    frmbytes = struct.unpack(">I", buf[pos + 4:pos + 8])[0]
    pfb = buf[pos + 8]
    f = dict(frmbytes=frmbytes, profile=pfb >> 5, ecc=bool(pfb >> 4 & 1), little_endian=bool(pfb >> 3 & 1),
             depth_idx=pfb & 7, force_flush=False, overlap_ratio=0)
    if f["profile"] in COMPACT:
        css = struct.unpack(">H", buf[pos + 9:pos + 11])[0]
        f.update(channels=(css >> 10) + 1, srate=COMPACT_SRATES[css >> 6 & 15],
                 fsize=COMPACT_SAMPLES[css >> 1 & 31], force_flush=bool(css & 1))
        if f["force_flush"]:
            return f, 12
        r = buf[pos + 11]
        f["overlap_ratio"] = r + 1 if r else 0
        hlen = 16 if f["ecc"] else 12
    else:
        f.update(channels=buf[pos + 9] + 1, srate=struct.unpack(">I", buf[pos + 12:pos + 16])[0],
                 fsize=struct.unpack(">I", buf[pos + 24:pos + 28])[0], crc=buf[pos + 28:pos + 32])
        hlen = 32
    if frmbytes == 0xFFFFFFFF:
        f["frmbytes"] = struct.unpack(">Q", buf[pos + hlen:pos + hlen + 8])[0]
        hlen += 8
    return f, hlen


def decode_stream(stream: bytes) -> np.ndarray:
    """Whole-stream restatement of ``Decoder.process()`` + ``flush()`` for a well-formed,
    ECC-free stream with constant channels/srate (ref: decoder.py:51-114)."""
    ola, out, pos, channels = OverlapAdd(), [], 0, 1
    while True:
        pos = stream.find(FRM_SIGN, pos)
        if pos < 0 or pos + 9 > len(stream):
            break
        f, hlen = asfh_parse(stream, pos)
        pos += hlen
        channels = f["channels"]
        if f["force_flush"]:
            out.append(ola.flush().reshape(-1, channels))
            continue
        frad = stream[pos:pos + f["frmbytes"]]
        pos += f["frmbytes"]
        if f["profile"] == 1:
            pcm = p1_digital(frad, f["depth_idx"], channels, f["srate"], f["fsize"])
        elif f["profile"] == 4:
            pcm = p4_digital(frad, f["depth_idx"], channels, f["little_endian"])
        else:
            pcm = p0_digital(frad, f["depth_idx"], channels, f["little_endian"])
        out.append(ola.push(pcm, f["profile"] in COMPACT, f["overlap_ratio"]))
    out.append(ola.flush().reshape(-1, channels))
    return np.concatenate(out) if out else np.zeros((0, channels))
