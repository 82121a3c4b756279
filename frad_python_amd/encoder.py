"""Streaming encoder with the reference's API (src/libfrad/encoder.py): same constructor, setters,
``process(bytes) -> EncodeResult`` and ``flush()``, byte-identical streams -- but every ``process`` call
hands ALL the whole frames it holds to the HIP transform core in one batched launch instead of looping
one frame at a time through NumPy (encoder.py:60-105).  ASFH framing and CRC stay on the host."""
from __future__ import annotations

import struct
import sys
import zlib

import numpy as np

from .backend.pcmformat import ff_format_to_numpy_type
from .fourier import AVAILABLE, BIT_DEPTHS, SEGMAX, profiles
from .fourier.profiles import compact
from .tools.asfh import ASFH

_LOSSLESS_DEPTHS = (12, 16, 24, 32, 48, 64)
_P1_DEPTHS = (8, 12, 16, 24, 32, 48, 64)


_POOL = None


def _map_zlib(fn, items: list) -> list:
    """deflate / inflate of a batch's frames on a small thread pool: zlib releases the GIL, the frames are independent and
    the results are the bytes the serial loop would give (profile1.py:50, :59).  Work is handed out in runs of frames so
    that the pool's per-task overhead (tens of microseconds) does not exceed a frame's own cost."""
    global _POOL
    n = len(items)
    if n < 32:
        return [fn(b) for b in items]
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))
    workers = _POOL._max_workers
    run = max(8, -(-n // (4 * workers)))
    chunks = [items[i:i + run] for i in range(0, n, run)]
    out = []
    for part in _POOL.map(lambda ch: [fn(b) for b in ch], chunks):
        out.extend(part)
    return out


class EncodeResult:
    def __init__(self, buf: bytes, samples: int):
        self.buf = buf
        self.samples = samples


class Encoder:
    def __init__(self, profile: int, srate: int, channels: int, bit_depth: int, frame_size: int, pcm_format: str, *, bridge=None):
        if profile not in AVAILABLE:
            print(f"Invalid profile! Available: {AVAILABLE}", file=sys.stderr)
            sys.exit(1)
        self.asfh = ASFH()
        self.buffer = b""
        self.bit_depth = self.channels = self.fsize = self.srate = 0
        self.pcm_format_name = pcm_format
        self.pcm_format = ff_format_to_numpy_type(pcm_format)
        self.loss_level = 0.5
        self.init = False
        self.have_carry = False          # compact profiles: the head of `buffer` is the previous frame's tail
        self._bridge = bridge
        self.set_profile(profile, srate, channels, bit_depth, frame_size)

    # ------------------------------------------------------------------ device access
    @property
    def bridge(self):
        if self._bridge is None:
            from .bridge import HipBridge
            self._bridge = HipBridge()
        return self._bridge

    # ------------------------------------------------------------------ framing helpers
    def _emit(self, frad: bytes, depth_idx: int, fsize: int) -> bytes:
        if self.asfh.ecc:
            raise NotImplementedError("Reed-Solomon ECC is host-side and not part of the MI355X transform core")
        a = self.asfh
        a.bit_depth_index, a.channels, a.fsize = depth_idx, self.channels, fsize
        a.srate = compact.get_valid_srate(self.srate) if a.profile in profiles.COMPACT else self.srate
        return a.write(frad)

    @staticmethod
    def _deflate(body: bytes) -> bytes:
        co = zlib.compressobj(zlib.Z_DEFAULT_COMPRESSION, zlib.DEFLATED, -15)     # profile1.py:50 wbits=-15
        return co.compress(body) + co.flush()

    def _encode_frames(self, pcm: bytes, n_frames: int, n_eff: int, hop: int, n_valid: int) -> bytes:
        """n_frames frames of n_eff sample-frames, frame i starting i*hop sample-frames into `pcm`."""
        prof, C = self.asfh.profile, self.channels
        out = []
        if prof == 1:
            bits = self.bit_depth if self.bit_depth in _P1_DEPTHS else 16
            N = compact.get_samples_min_ge(n_eff)
            # quantiser and Exp-Golomb-Rice coder behind the bridge (on the device); the host only deflates (profile1.py:50) and frames
            bodies = self.bridge.p1_encode_bodies(pcm, self.pcm_format_name, n_frames, N, C, bits, compact.get_valid_srate(self.srate),
                                                  self.loss_level, hop, n_valid)
            for frad in _map_zlib(self._deflate, bodies):
                out.append(self._emit(frad, _P1_DEPTHS.index(bits), n_valid))
        else:
            bits = self.bit_depth if self.bit_depth in _LOSSLESS_DEPTHS else 16
            whole = getattr(self.bridge, "lossless_encode_stream", None)
            if whole is not None and not self.asfh.ecc and n_frames > 1:
                # headers and checksums on the device, one copy back (bridge.py); falls through when a frame escalates
                a = self.asfh
                a.bit_depth_index, a.channels, a.fsize, a.srate = _LOSSLESS_DEPTHS.index(bits), C, n_eff, self.srate
                got = whole(prof, pcm, self.pcm_format_name, n_frames, n_eff, C, bits, a.endian, a.lossless_head)
                if got is not None:
                    return got
            for frad, used in self.bridge.lossless_encode(prof, pcm, self.pcm_format_name, n_frames, n_eff, C, bits, self.asfh.endian):
                out.append(self._emit(frad, _LOSSLESS_DEPTHS.index(used), n_eff))
        return b"".join(out)

    def inner(self, stream: bytes, flush: bool) -> EncodeResult:
        """Transactional: `buffer` / `have_carry` change only after every launch of this call has succeeded, so a device
        error leaves the encoder exactly as it was before the call (nothing consumed, nothing half-emitted)."""
        if not self.init:
            self.buffer += stream
            return EncodeResult(b"", 0)
        buf, have_carry = self.buffer + stream, self.have_carry
        compact_prof = self.asfh.profile in profiles.COMPACT
        n_eff = compact.get_samples_min_ge(self.fsize) if compact_prof else self.fsize
        ratio = self.asfh.overlap_ratio
        cut = n_eff * (ratio - 1) // ratio if (compact_prof and ratio > 1) else n_eff      # encoder.py:48
        step = self.pcm_format.itemsize * self.channels
        have = len(buf) // step
        ret, samples = b"", 0
        # ---- whole frames (encoder.py:72-104, batched): frame i covers sample-frames [i*cut, i*cut + n_eff)
        if have >= n_eff:
            k = (have - n_eff) // cut + 1
            ret += self._encode_frames(buf[:((k - 1) * cut + n_eff) * step], k, n_eff, cut, n_eff)
            carry = n_eff - cut
            samples += k * n_eff - (k - 1) * carry - (carry if have_carry else 0)
            buf = buf[k * cut * step:]
            have_carry = carry > 0
        if flush:
            # ---- flush (encoder.py:81, 89-91, 105): what is left (carry + remainder) becomes one short frame
            have = len(buf) // step
            if have > 0:
                carry = (n_eff - cut) if have_carry else 0
                ret += self._encode_frames(buf[:have * step], 1, have, have, have)
                samples += have - carry
                ret += self.asfh.force_flush()
            buf, have_carry = b"", False
            ret += self.asfh.force_flush()
        self.buffer, self.have_carry = buf, have_carry
        return EncodeResult(ret, samples)

    def process(self, stream: bytes) -> EncodeResult:
        return self.inner(stream, False)

    def flush(self) -> EncodeResult:
        return self.inner(b"", True) if self.init else EncodeResult(b"", 0)

    # ------------------------------------------------------------------ getters / setters (encoder.py:114-215)
    @staticmethod
    def verify_profile(profile):
        return None if profile in AVAILABLE else f"Invalid profile! Available: {AVAILABLE}"

    @staticmethod
    def verify_srate(profile, srate):
        if srate == 0:
            return "Sample rate cannot be zero"
        if profile in profiles.COMPACT and compact.get_valid_srate(srate) != srate:
            return f"Invalid sample rate! Valid rates for profile {profile}: {compact.SRATES}"
        return None

    @staticmethod
    def verify_channels(profile, channels):
        return "Channel count cannot be zero" if channels == 0 else None

    @staticmethod
    def verify_bit_depth(profile, bit_depth):
        if bit_depth == 0:
            return "Bit depth cannot be zero"
        if bit_depth not in BIT_DEPTHS[profile]:
            return f"Invalid bit depth! Valid depths for profile {profile}: {list(BIT_DEPTHS[profile])}"
        return None

    @staticmethod
    def verify_frame_size(profile, frame_size):
        if frame_size == 0:
            return "Frame size cannot be zero"
        if frame_size > SEGMAX[profile]:
            return f"Samples per frame cannot exceed {SEGMAX[profile]}"
        return None

    def get_profile(self):
        return self.asfh.profile

    def set_profile(self, profile, srate, channels, bit_depth, frame_size):
        for e in (self.verify_profile(profile), self.verify_srate(profile, srate), self.verify_channels(profile, channels),
                  self.verify_bit_depth(profile, bit_depth), self.verify_frame_size(profile, frame_size)):
            if e is not None:
                return e
        res = EncodeResult(b"", 0)
        if (self.channels != 0 and self.channels != channels) or (self.srate != 0 and self.srate != srate):
            res = self.flush()
        self.asfh.profile = profile
        self.srate, self.channels, self.bit_depth, self.fsize = srate, channels, bit_depth, frame_size
        self.init = True
        return res

    def get_channels(self):
        return self.channels

    def set_channels(self, channels):
        res = EncodeResult(b"", 0)
        if self.channels != 0 and self.channels != channels:
            res = self.flush()
        self.channels = channels
        return res

    def get_srate(self):
        return self.srate

    def set_srate(self, srate):
        if e := self.verify_srate(self.get_profile(), srate):
            return e
        res = EncodeResult(b"", 0)
        if self.srate != 0 and self.srate != srate:
            res = self.flush()
        self.srate = srate
        return res

    def get_frame_size(self):
        return self.fsize

    def set_frame_size(self, frame_size):
        if e := self.verify_frame_size(self.get_profile(), frame_size):
            return e
        self.fsize = frame_size

    def get_bit_depth(self):
        return self.bit_depth

    def set_bit_depth(self, bit_depth):
        if e := self.verify_bit_depth(self.get_profile(), bit_depth):
            return e
        self.bit_depth = bit_depth

    def set_ecc(self, ecc: bool, ecc_ratio):
        if ecc:
            raise NotImplementedError("Reed-Solomon ECC is host-side and not part of the MI355X transform core "
                                      "(the reference needs the third-party reedsolo module for it)")
        self.asfh.ecc = False
        self.asfh.ecc_dsize, self.asfh.ecc_codesize = ecc_ratio if ecc_ratio[0] and sum(ecc_ratio) <= 255 else (96, 24)

    def set_little_endian(self, little_endian: bool):
        self.asfh.endian = little_endian

    def set_loss_level(self, loss_level: float):
        self.loss_level = max(abs(loss_level), 0.125)

    def set_overlap_ratio(self, overlap_ratio: int):
        if overlap_ratio != 0:
            overlap_ratio = max(2, min(256, overlap_ratio))
        self.asfh.overlap_ratio = overlap_ratio
