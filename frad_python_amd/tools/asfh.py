"""Audio Stream Frame Header -- host framing (mirror of src/libfrad/tools/asfh.py of the reference).

Layout (big-endian): FRM_SIGN(4) len(4) pfb(1) then
  lossless profiles: channels-1 (1) ecc dsize,codesize (2) srate (4) zero (8) fsize (4) crc32 (4) = 32 bytes
  compact profiles : css (2) overlap-1 (1) [ecc dsize,codesize (2) crc16 (2)]                    = 12 / 16 bytes
pfb = profile<<5 | ecc<<4 | little_endian<<3 | depth index; css = (ch-1)<<10 | srate idx<<6 | fsize idx<<1 | flush.
Framing, CRC and Reed-Solomon stay on the host CPU by design (BASELINE.json north_star)."""
from __future__ import annotations

import struct
from zlib import crc32

from ..common import FRM_SIGN, crc16_ansi
from ..fourier.profiles import COMPACT, compact


def encode_pfb(profile: int, isecc: bool, little_endian: bool, bits: int) -> bytes:
    return struct.pack("<B", (profile << 5) | (int(bool(isecc)) << 4) | (int(bool(little_endian)) << 3) | bits)


def decode_pfb(pfb: bytes):
    v = pfb[0]
    return v >> 5, bool(v >> 4 & 1), bool(v >> 3 & 1), v & 7


def encode_css_prf1(channels: int, srate: int, fsize: int, force_flush: bool) -> bytes:
    return struct.pack(">H", ((channels - 1) << 10) | (compact.get_srate_index(srate) << 6)
                       | (compact.get_samples_index(fsize) << 1) | int(bool(force_flush)))


def decode_css_prf1(css: bytes):
    v = struct.unpack(">H", css)[0]
    return (v >> 10) + 1, compact.SRATES[v >> 6 & 15], compact.SAMPLES[v >> 1 & 31], bool(v & 1)


class ASFH:
    def __init__(self):
        self.frmbytes, self.buffer, self.all_set, self.header_bytes = 0, b"", False, 0
        self.endian, self.bit_depth_index = False, 0
        self.channels, self.srate, self.fsize = 0, 0, 0
        self.ecc, self.ecc_dsize, self.ecc_codesize = False, 0, 0
        self.profile, self.overlap_ratio = 0, 0
        self.crc = b""

    def criteq(self, other: "ASFH") -> bool:
        return self.channels == other.channels and self.srate == other.srate

    def write(self, frad: bytes) -> bytes:
        head = FRM_SIGN + struct.pack(">I", len(frad)) + encode_pfb(self.profile, self.ecc, self.endian, self.bit_depth_index)
        if self.profile in COMPACT:
            head += encode_css_prf1(self.channels, self.srate, self.fsize, False)
            head += struct.pack("B", max(self.overlap_ratio - 1, 0))
            if self.ecc:
                head += struct.pack("BB", self.ecc_dsize, self.ecc_codesize) + crc16_ansi(frad).to_bytes(2, "big")
        else:
            head += struct.pack("B", self.channels - 1) + struct.pack("BB", self.ecc_dsize, self.ecc_codesize)
            head += struct.pack(">I", self.srate) + b"\x00" * 8 + struct.pack(">I", self.fsize)
            head += crc32(frad).to_bytes(4, "big")
        return head + frad

    def force_flush(self) -> bytes:
        if self.profile not in COMPACT:
            return b""
        head = FRM_SIGN + b"\x00" * 4 + encode_pfb(self.profile, self.ecc, self.endian, self.bit_depth_index)
        return head + encode_css_prf1(max(self.channels, 1), self.srate, self.fsize, True) + b"\x00"

    def fill_buffer(self, buffer: bytes, target: int):
        if len(self.buffer) < target:
            cut = target - len(self.buffer)
            self.buffer += buffer[:cut]
            buffer = buffer[cut:]
            if len(self.buffer) < target:
                return False, buffer
        self.header_bytes = target
        return True, buffer

    def read(self, buffer: bytes):
        ok, buffer = self.fill_buffer(buffer, 9)
        if not ok:
            return "Incomplete", buffer
        self.frmbytes = struct.unpack(">I", self.buffer[4:8])[0]
        self.profile, self.ecc, self.endian, self.bit_depth_index = decode_pfb(self.buffer[8:9])
        if self.profile in COMPACT:
            ok, buffer = self.fill_buffer(buffer, 12)
            if not ok:
                return "Incomplete", buffer
            self.channels, self.srate, self.fsize, flush = decode_css_prf1(self.buffer[9:11])
            if flush:
                return "ForceFlush", buffer
            self.overlap_ratio = self.buffer[11]
            if self.overlap_ratio != 0:
                self.overlap_ratio += 1
            if self.ecc:
                ok, buffer = self.fill_buffer(buffer, 16)
                if not ok:
                    return "Incomplete", buffer
                self.ecc_dsize, self.ecc_codesize = struct.unpack("BB", self.buffer[12:14])
                self.crc = self.buffer[14:16]
        else:
            ok, buffer = self.fill_buffer(buffer, 32)
            if not ok:
                return "Incomplete", buffer
            self.channels = self.buffer[9] + 1
            self.ecc_dsize, self.ecc_codesize = struct.unpack("BB", self.buffer[10:12])
            self.srate = struct.unpack(">I", self.buffer[12:16])[0]
            self.fsize = struct.unpack(">I", self.buffer[24:28])[0]
            self.crc = self.buffer[28:32]
        if self.frmbytes == 0xFFFFFFFF:
            ok, buffer = self.fill_buffer(buffer, self.header_bytes + 8)
            if not ok:
                return "Incomplete", buffer
            self.frmbytes = struct.unpack(">Q", self.buffer[-8:])[0]
        self.all_set = True
        return "Complete", buffer

    def clear(self):
        self.all_set, self.buffer = False, b""
