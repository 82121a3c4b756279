"""Audio Stream Frame Header (ASFH) -- host framing of the FrAD stream.

Byte-compatible with the reference's `ASFH` (src/libfrad/tools/asfh.py) and same public surface
(`write`, `read` -> 'Complete' | 'ForceFlush' | 'Incomplete', `force_flush`, `clear`, `criteq`, the public
fields), because Encoder/Decoder and their callers use it as the frame descriptor.  Layout, big-endian:

    0  FRM_SIGN(4) | 4 payload length(4) | 8 pfb(1) = profile<<5 | ecc<<4 | little_endian<<3 | depth index
    lossless profiles (0, 4), 32 bytes:  9 channels-1 | 10 ecc dsize | 11 ecc codesize | 12 srate(4) | 16 zero(8)
                                         | 24 fsize(4) | 28 crc32(payload)(4)
    compact profiles (1, 2), 12 bytes :  9 css(2) = (channels-1)<<10 | srate idx<<6 | fsize idx<<1 | flush
                                         | 11 overlap_ratio-1 | with ECC, 16 bytes: 12 dsize | 13 codesize | 14 crc16(2)
    length field 0xFFFFFFFF: a u64 length follows the header.
Framing, CRC and Reed-Solomon stay on the host CPU by design (BASELINE.json north_star)."""
from __future__ import annotations

import struct
from zlib import crc32

from ..common import FRM_SIGN, crc16_ansi
from ..fourier.profiles import COMPACT, compact

_LOSSLESS_TAIL = struct.Struct(">BBBI8xI")       # channels-1, dsize, codesize, srate, 8 zero bytes, fsize
_HEAD_MIN, _HEAD_COMPACT, _HEAD_COMPACT_ECC, _HEAD_LOSSLESS = 9, 12, 16, 32


def encode_pfb(profile: int, isecc: bool, little_endian: bool, bits: int) -> bytes:
    return bytes([(profile << 5) | (bool(isecc) << 4) | (bool(little_endian) << 3) | bits])


def decode_pfb(pfb: bytes):
    v = pfb[0]
    return v >> 5, bool(v & 0x10), bool(v & 0x08), v & 0x07


def encode_css_prf1(channels: int, srate: int, fsize: int, force_flush: bool) -> bytes:
    word = ((channels - 1) << 10) | (compact.get_srate_index(srate) << 6) | (compact.get_samples_index(fsize) << 1)
    return (word | bool(force_flush)).to_bytes(2, "big")


def decode_css_prf1(css: bytes):
    word = int.from_bytes(css, "big")
    return (word >> 10) + 1, compact.SRATES[(word >> 6) & 0xF], compact.SAMPLES[(word >> 1) & 0x1F], bool(word & 1)


class ASFH:
    def __init__(self):
        self.frmbytes = 0
        self.buffer = b""                 # header bytes gathered so far (across `read` calls)
        self.all_set = False
        self.header_bytes = 0
        self.endian = False
        self.bit_depth_index = 0
        self.channels = self.srate = self.fsize = 0
        self.ecc = False
        self.ecc_dsize = self.ecc_codesize = 0
        self.profile = 0
        self.overlap_ratio = 0
        self.crc = b""

    def criteq(self, other: "ASFH") -> bool:
        return (self.channels, self.srate) == (other.channels, other.srate)

    # ------------------------------------------------------------------ writing
    def _prefix(self, length: int) -> bytes:
        return FRM_SIGN + length.to_bytes(4, "big") + encode_pfb(self.profile, self.ecc, self.endian, self.bit_depth_index)

    def lossless_head(self, length: int) -> bytes:
        """The 28 header bytes of a lossless frame that do not depend on its payload (the CRC-32 follows)."""
        return self._prefix(length) + _LOSSLESS_TAIL.pack(self.channels - 1, self.ecc_dsize, self.ecc_codesize, self.srate, self.fsize)

    def write(self, frad: bytes) -> bytes:
        parts = [self._prefix(len(frad))]
        if self.profile in COMPACT:
            parts.append(encode_css_prf1(self.channels, self.srate, self.fsize, False))
            parts.append(bytes([max(self.overlap_ratio - 1, 0)]))
            if self.ecc:
                parts.append(bytes([self.ecc_dsize, self.ecc_codesize]) + crc16_ansi(frad).to_bytes(2, "big"))
        else:
            parts.append(_LOSSLESS_TAIL.pack(self.channels - 1, self.ecc_dsize, self.ecc_codesize, self.srate, self.fsize))
            parts.append(crc32(frad).to_bytes(4, "big"))
        parts.append(frad)
        return b"".join(parts)

    def force_flush(self) -> bytes:
        if self.profile not in COMPACT:
            return b""
        return self._prefix(0) + encode_css_prf1(max(self.channels, 1), self.srate, self.fsize, True) + b"\x00"

    # ------------------------------------------------------------------ reading (incremental)
    def fill_buffer(self, buffer: bytes, target_size: int):
        """Move bytes from `buffer` into the header until it holds `target_size`; (done, rest)."""
        missing = target_size - len(self.buffer)
        if missing > 0:
            self.buffer += buffer[:missing]
            buffer = buffer[missing:]
        done = len(self.buffer) >= target_size
        if done:
            self.header_bytes = target_size
        return done, buffer

    def read(self, buffer: bytes):
        done, buffer = self.fill_buffer(buffer, _HEAD_MIN)
        if not done:
            return "Incomplete", buffer
        h = self.buffer
        self.frmbytes = int.from_bytes(h[4:8], "big")
        self.profile, self.ecc, self.endian, self.bit_depth_index = decode_pfb(h[8:9])
        compact_profile = self.profile in COMPACT
        done, buffer = self.fill_buffer(buffer, _HEAD_COMPACT if compact_profile else _HEAD_LOSSLESS)
        if not done:
            return "Incomplete", buffer
        h = self.buffer
        if compact_profile:
            self.channels, self.srate, self.fsize, flush = decode_css_prf1(h[9:11])
            if flush:
                return "ForceFlush", buffer
            self.overlap_ratio = h[11] + 1 if h[11] else 0
            if self.ecc:
                done, buffer = self.fill_buffer(buffer, _HEAD_COMPACT_ECC)
                if not done:
                    return "Incomplete", buffer
                h = self.buffer
                self.ecc_dsize, self.ecc_codesize, self.crc = h[12], h[13], h[14:16]
        else:
            ch1, self.ecc_dsize, self.ecc_codesize, self.srate, self.fsize = _LOSSLESS_TAIL.unpack(h[9:28])
            self.channels, self.crc = ch1 + 1, h[28:32]
        if self.frmbytes == 0xFFFFFFFF:                      # 64-bit length extension
            done, buffer = self.fill_buffer(buffer, self.header_bytes + 8)
            if not done:
                return "Incomplete", buffer
            self.frmbytes = int.from_bytes(self.buffer[-8:], "big")
        self.all_set = True
        return "Complete", buffer

    def clear(self):
        self.all_set = False
        self.buffer = b""
