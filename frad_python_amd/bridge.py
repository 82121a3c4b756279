"""Host <-> device bridge used by the streaming Encoder / Decoder.

``HipBridge`` moves a batch of frames (host bytes / ndarrays, as the reference's streaming API hands
them over) through the HIP transform core: one H2D copy, one launch per homogeneous batch, one D2H
copy.  The Encoder / Decoder only depend on this small interface, which lets the CPU-only test-suite
drive their host logic (frame cut, overlap carry, ASFH, CRC) with a bridge of its own; the default
-- and the only one in this package -- is the HIP one: there is no CPU fallback."""
from __future__ import annotations

import numpy as np


class HipBridge:
    _pinned = None

    def __init__(self, device=None):
        import torch
        from . import core
        if not torch.cuda.is_available():
            raise RuntimeError("the FrAD transform core needs an MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.torch, self.core = torch, core
        self.scan_lib = core._lib.load()                      # frad_asfh_scan: the decoder's native header scanner
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def _up(self, data: bytes):
        """host bytes -> device uint8 tensor, straight from the caller's (read-only) buffer"""
        import warnings
        if not len(data):
            return self.torch.empty(0, dtype=self.torch.uint8, device=self.device)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                   # torch warns about non-writable memory; it is only read
            return self.torch.frombuffer(data, dtype=self.torch.uint8).to(self.device)

    def _down_bytes(self, dev) -> bytes:
        """device uint8 tensor -> bytes through a cached pinned staging buffer (no pageable bounce, no fresh
        page faults for the intermediate copy)"""
        t = self.torch
        n = dev.numel()
        pin = HipBridge._pinned                               # one staging buffer per process, grown on demand
        if pin is None or pin.numel() < n:
            pin = HipBridge._pinned = t.empty(max(n, 1 << 20), dtype=t.uint8, pin_memory=True)
        pin[:n].copy_(dev.reshape(-1))
        return pin[:n].numpy().tobytes()

    def lossless_encode(self, profile, pcm: bytes, fmt, n_frames, N, C, bits, little_endian, raw_be_ints=True):
        """-> list of (payload bytes, bits actually used) per frame."""
        enc = self.core.analogue_batch(profile, self._up(pcm), fmt, n_frames, N, C, bits, little_endian,
                                       raw_be_ints=raw_be_ints)
        host = enc.payload[:, :enc.nbytes].cpu().numpy()
        out = []
        for i in range(n_frames):
            if i in enc.escalated:
                row, b = enc.escalated[i]
                out.append((bytes(row.cpu().numpy()), b))
            else:
                out.append((host[i].tobytes(), enc.bits))
        return out

    def lossless_encode_stream(self, profile, pcm: bytes, fmt, n_frames, N, C, bits, little_endian, head_fn,
                               raw_be_ints=True):
        """Whole batch -> finished stream bytes, assembled on the device: the payload kernel writes every
        frame behind a 32-byte hole, ``frad_crc32_frames`` fills in the checksums, the constant part of the
        header (``head_fn(payload_bytes)`` -> 28 bytes, tools/asfh.py) is broadcast, and one D2H copy returns
        the result.  None when a frame needs a deeper format or a 64-bit length: the caller then goes frame
        by frame through ``lossless_encode``."""
        t, core = self.torch, self.core
        if bits not in core.DEPTHS:
            bits = 16
        nb = core._lib.load().payload_bytes(N, C, bits)
        if n_frames == 0 or nb >= 0xFFFFFFFF:
            return None
        stream = t.empty((n_frames, 32 + nb), dtype=t.uint8, device=self.device)
        pay = stream[:, 32:]
        enc = core.analogue_batch(profile, self._up(pcm), fmt, n_frames, N, C, bits, little_endian,
                                  raw_be_ints=raw_be_ints, out=pay)
        if enc.escalated:
            return None
        crc = core.crc32_frames(pay, nb)
        stream[:, :28] = t.frombuffer(bytearray(head_fn(nb)), dtype=t.uint8).to(self.device)
        stream[:, 28:32] = crc.view(t.uint8).view(n_frames, 4).flip(1)       # big-endian, as int.to_bytes(4, "big")
        return self._down_bytes(stream)

    def _down_array(self, dev, dtype, shape) -> np.ndarray:
        """device tensor -> numpy array backed by a pinned host block of its own (torch's caching host allocator hands
        the block out again once the array is gone): one DMA, no pageable bounce, no second host copy"""
        t = self.torch
        host = t.empty(dev.numel() * dev.element_size(), dtype=t.uint8, pin_memory=True)
        host.copy_(dev.reshape(-1).view(t.uint8), non_blocking=True)
        t.cuda.current_stream(self.device).synchronize()
        return host.numpy().view(dtype).reshape(shape)

    def lossless_decode_strided(self, profile, region, n_frames, stride, nbytes, N, C, bits, little_endian, out_format=None) -> np.ndarray:
        """Frames that sit equally spaced in the stream (``region`` = first payload byte .. last payload byte, a
        read-only buffer): one H2D copy of the region, headers and all, and the kernels step over it with
        ``payload_stride = stride``; no per-frame host copies."""
        import warnings
        t = self.torch
        if nbytes != self.core._lib.load().payload_bytes(N, C, bits):
            return None                                       # header length and geometry disagree: frame-by-frame path decides
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                   # read-only buffer: it is only copied to the device
            dev = t.frombuffer(region, dtype=t.uint8).to(self.device)
        if out_format is not None:                            # narrowed on the device: 2-4x fewer bytes over PCIe
            from .backend.pcmformat import ff_format_to_numpy_type
            out = self.core.digital_batch(profile, dev, n_frames, N, C, bits, little_endian, payload_stride=stride, out_format=out_format)
            return self._down_array(out, ff_format_to_numpy_type(out_format), (n_frames, N, C))
        out = self.core.digital_batch(profile, dev, n_frames, N, C, bits, little_endian, payload_stride=stride)
        return self._down_array(out, np.float64, (n_frames, N, C))

    def lossless_decode(self, profile, payloads: list, N, C, bits, little_endian) -> np.ndarray:
        n = len(payloads)
        nb = len(payloads[0])
        stride = (nb + 15) // 16 * 16
        host = np.zeros((n, stride), np.uint8)
        for i, p in enumerate(payloads):
            host[i, :nb] = np.frombuffer(p, np.uint8)
        dev = self.torch.from_numpy(host).to(self.device)
        return self.core.digital_batch(profile, dev, n, N, C, bits, little_endian).cpu().numpy()

    def p1_encode(self, pcm: bytes, fmt, n_frames, N, C, bits, srate, loss_level, hop, n_valid, raw_be_ints=True):
        q, tq = self.core.p1_analogue_batch(self._up(pcm), fmt, n_frames, N, C, bits, srate, loss_level,
                                            frame_stride=hop, n_valid=n_valid, raw_be_ints=raw_be_ints)
        return q.cpu().numpy(), tq.cpu().numpy()

    def p1_encode_bodies(self, pcm: bytes, fmt, n_frames, N, C, bits, srate, loss_level, hop, n_valid, raw_be_ints=True) -> list:
        """K7 + the Exp-Golomb-Rice stage on the device: the pre-deflate body of every frame (profile1.py:15-45),
        one D2H copy of exactly those bytes; the integers never leave the device."""
        q, tq = self.core.p1_analogue_batch(self._up(pcm), fmt, n_frames, N, C, bits, srate, loss_level,
                                            frame_stride=hop, n_valid=n_valid, raw_be_ints=raw_be_ints)
        flat, offsets = self.core.p1_golomb_encode_batch(q, tq)
        off = offsets.cpu().numpy()
        host = self._down_bytes(flat) if flat.numel() else b""
        return [host[off[i]:off[i + 1]] for i in range(n_frames)]

    def p1_decode_bodies(self, bodies: list, N, C, bits, srate) -> np.ndarray:
        """Inflated frame bodies -> PCM: Golomb decode (profile1.py:59-64) and K8 on the device, one H2D copy of the
        bodies (about a byte per coefficient instead of the four of an int32 array)."""
        t = self.torch
        off = np.zeros(len(bodies) + 1, np.int64)
        np.cumsum([len(b) for b in bodies], out=off[1:])
        flat = self._up(b"".join(bodies) + bytes(8))              # the decoder reads the stream as aligned 32-bit words: tail slack (frad_hip.h)
        q, tq, status = self.core.p1_golomb_decode_batch(flat, t.from_numpy(off).to(self.device), N, C)
        return self.core.p1_digital_batch(q, tq, N, C, bits, srate).cpu().numpy()

    def p1_decode_run(self, bodies: list, N, C, bits, srate, ratio, prev_tail, out_format=None):
        """A run of overlapped compact frames, whole on the device: Golomb decode, K8, the Hann cross-fade against
        ``prev_tail`` (decoder.py:28-46) and -- with ``out_format`` -- the output conversion in the same pass.  One upload
        of the inflated bodies, one download of the finished PCM ``[n_frames * cut, C]`` and of the new tail (float64)."""
        from .backend.pcmformat import ff_format_to_numpy_type
        t = self.torch
        off = np.zeros(len(bodies) + 1, np.int64)
        np.cumsum([len(b) for b in bodies], out=off[1:])
        flat = self._up(b"".join(bodies) + bytes(8))
        q, tq, status = self.core.p1_golomb_decode_batch(flat, t.from_numpy(off).to(self.device), N, C)
        frames = self.core.p1_digital_batch(q, tq, N, C, bits, srate)
        pt = t.from_numpy(np.ascontiguousarray(prev_tail)).to(self.device) if prev_tail is not None else None
        out, nxt = self.core.p1_overlap_add(frames, ratio, pt, out_format=out_format)
        if out.dtype == t.uint8:
            pcm = np.frombuffer(out.cpu().numpy().tobytes(), ff_format_to_numpy_type(out_format)).reshape(-1, C)
        else:
            pcm = out.cpu().numpy().reshape(-1, C)
        return pcm, nxt.cpu().numpy()

    def p1_decode(self, q: np.ndarray, tq: np.ndarray, N, C, bits, srate) -> np.ndarray:
        t = self.torch
        return self.core.p1_digital_batch(t.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.device),
                                          t.from_numpy(np.ascontiguousarray(tq, np.int32)).to(self.device),
                                          N, C, bits, srate).cpu().numpy()

    def overlap_add(self, frames: np.ndarray, ratio: int, prev_tail):
        t = self.torch
        pt = t.from_numpy(np.ascontiguousarray(prev_tail)).to(self.device) if prev_tail is not None else None
        out, nxt = self.core.p1_overlap_add(t.from_numpy(np.ascontiguousarray(frames)).to(self.device), ratio, pt)
        return out.cpu().numpy(), nxt.cpu().numpy()
