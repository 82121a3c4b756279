"""Shared plumbing of the parity tests.

Two ways to reach the kernels, same C-ABI either way:
  * ``GpuBackend``  -- the product path: frad_python_amd.core on cuda:0 (libfrad_hip.so, gfx950);
  * ``EmuBackend``  -- tests/emu: the same kernel source interpreted on the CPU (no GPU needed),
                       used by the ``-m "not gpu"`` suite to check indexing/packing logic.
"""
from __future__ import annotations

import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.environ.get("FRAD_EMU_LIB") or os.path.join(EMU_DIR, "libfrad_emu.so")   # FRAD_EMU_LIB: the ASan build (make emu-asan)
CSRC = os.path.join(ROOT, "frad_python_amd", "csrc")


def build_emulator() -> str:
    """make emu (g++, thread-per-lane interpreter of the same kernel source); no-op when current."""
    if not os.environ.get("FRAD_EMU_LIB"):
        subprocess.run(["make", "-j8", "emu"], check=True, cwd=CSRC, stdout=subprocess.DEVNULL)
    return EMU_LIB


def _align16(n):
    return (n + 15) // 16 * 16


class EmuBackend:
    name = "emu"

    def __init__(self):
        from frad_python_amd._lib import FradLib
        self.lib = FradLib(build_emulator())

    def analogue(self, profile, raw: np.ndarray, fmt, F, N, C, bits, le, frame_stride=None, raw_be=True,
                 offset=0, pad_stride=0):
        """raw: contiguous ndarray of PCM elements.  ``offset`` (bytes) mis-aligns both buffers."""
        from frad_python_amd.backend.pcmformat import pcm_dtype_code
        pb = self.lib.payload_bytes(N, C, bits if bits in (12, 16, 24, 32, 48, 64) else 16)
        stride = (_align16(pb) if not offset else pb) + pad_stride
        src = np.zeros(raw.nbytes + offset + 64, np.uint8)
        src[offset:offset + raw.nbytes] = raw.view(np.uint8).reshape(-1)
        pay = np.zeros(F * stride + offset + 64, np.uint8)
        am = np.zeros(max(F, 1))
        fn = self.lib.p4_analogue if profile == 4 else self.lib.p0_analogue
        flags = int(le) | (2 if raw_be else 0)
        fn(src.ctypes.data + offset, pcm_dtype_code(fmt), F, N, C, N if frame_stride is None else frame_stride, bits,
           flags, pay.ctypes.data + offset, stride, am.ctypes.data)
        return pay[offset:offset + F * stride].reshape(F, stride)[:, :pb].copy(), am[:F]

    def digital(self, profile, payload: np.ndarray, F, N, C, bits, le, offset=0):
        pb = payload.shape[1] if F else 0
        stride = _align16(pb) if not offset else pb
        buf = np.zeros(F * stride + offset + 64, np.uint8)
        if F:
            buf[offset:offset + F * stride].reshape(F, stride)[:, :pb] = payload
        out = np.zeros((max(F, 1), N, C))
        fn = self.lib.p4_digital if profile == 4 else self.lib.p0_digital
        fn(buf.ctypes.data + offset, stride, F, N, C, bits, int(le), out.ctypes.data)
        return out[:F]

    def clips(self, raw: np.ndarray, fmt, N, bits, first=0, fpc=None):
        """frad_p0_analogue_clips + frad_p0_digital_clips over raw [n_clips, clip_len, C] -> (payload rows, absmax, decoded clips)"""
        from frad_python_amd.backend.pcmformat import pcm_dtype_code
        n_clips, clip_len, C = raw.shape
        fpc = (clip_len - first) // N if fpc is None else fpc
        pb = self.lib.payload_bytes(N, C, bits); stride = _align16(pb)
        src = np.zeros(raw.nbytes + 64, np.uint8); src[:raw.nbytes] = raw.view(np.uint8).reshape(-1)
        F = n_clips * fpc
        pay = np.zeros(F * stride + 64, np.uint8); am = np.zeros(F)
        flag = np.zeros(1, np.int32)
        self.lib.p0_analogue_clips(src.ctypes.data + first * C * raw.itemsize, pcm_dtype_code(fmt), n_clips, clip_len, fpc, N, C, bits, 2,
                                   pay.ctypes.data, stride, am.ctypes.data, flag.ctypes.data)
        out = np.full((n_clips, clip_len, C), -7.0)
        self.lib.p0_digital_clips(pay.ctypes.data, stride, n_clips, fpc, N, C, bits, 0, out.ctypes.data + first * C * 8, clip_len)
        return pay[:F * stride].reshape(F, stride)[:, :pb].copy(), am, out

    def crc32_frames(self, rows: np.ndarray, nbytes, offset=0):
        F, stride = rows.shape
        buf = np.zeros(F * stride + offset + 64, np.uint8)
        buf[offset:offset + F * stride] = rows.reshape(-1)
        out = np.zeros(max(F, 1), np.uint32)
        self.lib.crc32_frames(buf.ctypes.data + offset, stride, F, nbytes, out.ctypes.data)
        return out[:F]

    def overflow_scan(self, absmax, bits, flag0=0):
        am = np.ascontiguousarray(absmax, np.float64)
        flag = np.array([flag0], np.int32)
        self.lib.p0_overflow_scan(am.ctypes.data, am.size, bits, flag.ctypes.data)
        return int(flag[0])

    # ---- profile 1 ----
    def p1_analogue(self, raw, fmt, F, N, C, bits, srate, loss, frame_stride=None, n_valid=None):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code
        src = np.zeros(raw.nbytes + 64, np.uint8); src[:raw.nbytes] = raw.view(np.uint8).reshape(-1)
        q = np.zeros((max(F, 1), N, C), np.int32); tq = np.zeros((max(F, 1), 27, C), np.int32)
        self.lib.p1_analogue(src.ctypes.data, pcm_dtype_code(fmt), F, N, C, N if frame_stride is None else frame_stride,
                             N if n_valid is None else n_valid, bits, srate, loss, 2, q.ctypes.data, tq.ctypes.data)
        return q[:F], tq[:F]

    def p1_digital(self, q, tq, N, C, bits, srate):
        F = q.shape[0]
        out = np.zeros((max(F, 1), N, C))
        qq, tt = np.ascontiguousarray(q, np.int32), np.ascontiguousarray(tq, np.int32)
        self.lib.p1_digital(qq.ctypes.data, tt.ctypes.data, F, N, C, bits, srate, out.ctypes.data)
        return out[:F]

    def p1_digital_pcm(self, q, tq, N, C, bits, srate, fmt):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code, ff_format_to_numpy_type
        F = q.shape[0]
        dt = ff_format_to_numpy_type(fmt)
        out = np.zeros(F * N * C * dt.itemsize + 16, np.uint8)
        qq, tt = np.ascontiguousarray(q, np.int32), np.ascontiguousarray(tq, np.int32)
        self.lib.p1_digital_pcm(qq.ctypes.data, tt.ctypes.data, F, N, C, bits, srate, pcm_dtype_code(fmt), out.ctypes.data)
        return np.frombuffer(out[:F * N * C * dt.itemsize].tobytes(), dt).reshape(F, N, C)

    def from_f64(self, x, fmt):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code, ff_format_to_numpy_type
        x = np.ascontiguousarray(x, np.float64)
        dt = ff_format_to_numpy_type(fmt)
        out = np.zeros(x.size * dt.itemsize + 16, np.uint8)
        self.lib.from_f64(x.ctypes.data, x.size, pcm_dtype_code(fmt), out.ctypes.data)
        return np.frombuffer(out[:x.size * dt.itemsize].tobytes(), dt)

    def digital_pcm(self, profile, payload: np.ndarray, F, N, C, bits, le, fmt):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code, ff_format_to_numpy_type
        dt = ff_format_to_numpy_type(fmt)
        pb = payload.shape[1]
        stride = _align16(pb)
        buf = np.zeros(F * stride + 64, np.uint8)
        buf[:F * stride].reshape(F, stride)[:, :pb] = payload
        out = np.zeros(F * N * C * dt.itemsize + 16, np.uint8)
        fn = self.lib.p4_digital_pcm if profile == 4 else self.lib.p0_digital_pcm
        fn(buf.ctypes.data, stride, F, N, C, bits, int(le) | 2, pcm_dtype_code(fmt), out.ctypes.data)
        return np.frombuffer(out[:F * N * C * dt.itemsize].tobytes(), dt).reshape(F, N, C)

    def golomb_encode(self, q, tq, offset=1):
        """-> list of per-frame pre-deflate bodies (frad_p1_golomb_encode + frad_rows_compact)"""
        q, tq = np.ascontiguousarray(q, np.int32), np.ascontiguousarray(tq, np.int32)
        F, N, C = q.shape
        stride = self.lib.p1_golomb_bound(N, C)
        rows = np.full(F * stride + 16, 0xAA, np.uint8); nb = np.zeros(max(F, 1), np.int64)
        base = rows.ctypes.data + (-rows.ctypes.data) % 16
        self.lib.p1_golomb_encode(q.ctypes.data, tq.ctypes.data, F, N, C, base, stride, nb.ctypes.data)
        offs = np.zeros(F + 1, np.int64); out = np.zeros(int(nb[:F].sum()) + 8 + offset, np.uint8)
        self.lib.rows_compact(base, stride, nb.ctypes.data, F, out.ctypes.data + offset, offs.ctypes.data)     # mis-aligned target
        return [bytes(out[offset + offs[i]:offset + offs[i + 1]]) for i in range(F)]

    def golomb_decode(self, bodies, N, C):
        F = len(bodies)
        offs = np.zeros(F + 1, np.int64); np.cumsum([len(b) for b in bodies], out=offs[1:])
        buf = np.frombuffer(b"".join(bodies) + b"\0" * 8, np.uint8).copy()
        q = np.full((max(F, 1), N, C), 7, np.int32); tq = np.full((max(F, 1), 27, C), 7, np.int32); st = np.zeros(max(F, 1), np.int32)
        self.lib.p1_golomb_decode(buf.ctypes.data, offs.ctypes.data, F, N, C, q.ctypes.data, tq.ctypes.data, st.ctypes.data)
        return q[:F], tq[:F], st[:F]

    def golomb_decode_fast_only(self, bodies, N, C, with_maps):
        """the wave-per-frame kernels alone (frad_debug_golomb_decode_wave): -> q, tq, todo (bit 0 / 1: stream left to the slow kernel)"""
        import ctypes as ct
        F = len(bodies)
        offs = np.zeros(F + 1, np.int64); np.cumsum([len(b) for b in bodies], out=offs[1:])
        buf = np.frombuffer(b"".join(bodies) + b"\0" * 8, np.uint8).copy()
        q = np.zeros((F, N, C), np.int32); tq = np.zeros((F, 27, C), np.int32); todo = np.full(F, -1, np.int32)
        rc = self.lib.dll.frad_debug_golomb_decode_wave(ct.c_void_p(buf.ctypes.data), ct.c_void_p(offs.ctypes.data), ct.c_int64(F), ct.c_int32(N), ct.c_int32(C),
                                                        ct.c_void_p(q.ctypes.data), ct.c_void_p(tq.ctypes.data), ct.c_void_p(todo.ctypes.data), ct.c_int32(with_maps), ct.c_void_p(0))
        assert rc == 0
        return q, tq, todo

    def p1_ola(self, frames, ratio, prev_tail=None):
        F, N, C = frames.shape
        cut = N * (ratio - 1) // ratio
        fr = np.ascontiguousarray(frames)
        out = np.zeros((F, cut, C)); nxt = np.zeros((N - cut, C))
        pt = np.ascontiguousarray(prev_tail) if prev_tail is not None else None
        self.lib.p1_overlap_add(fr.ctypes.data, F, N, C, ratio, pt.ctypes.data if pt is not None else 0, out.ctypes.data, nxt.ctypes.data)
        return out, nxt

    def p1_ola_pcm(self, frames, ratio, fmt, prev_tail=None):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code, ff_format_to_numpy_type
        F, N, C = frames.shape
        cut = N * (ratio - 1) // ratio
        dt = ff_format_to_numpy_type(fmt)
        fr = np.ascontiguousarray(frames)
        out = np.zeros(F * cut * C * dt.itemsize + 16, np.uint8); nxt = np.zeros((N - cut, C))
        pt = np.ascontiguousarray(prev_tail) if prev_tail is not None else None
        self.lib.p1_overlap_add_pcm(fr.ctypes.data, F, N, C, ratio, pt.ctypes.data if pt is not None else 0, pcm_dtype_code(fmt), out.ctypes.data, nxt.ctypes.data)
        return np.frombuffer(out[:F * cut * C * dt.itemsize].tobytes(), dt).reshape(F, cut, C), nxt


class GpuBackend:
    name = "gpu"

    def __init__(self):
        import torch
        assert torch.cuda.is_available()
        self.torch = torch
        self.dev = torch.device("cuda:0")

    def analogue(self, profile, raw, fmt, F, N, C, bits, le, frame_stride=None, raw_be=True, offset=0, pad_stride=0):
        from frad_python_amd import core
        t = self.torch
        src = t.zeros(raw.nbytes + offset + 64, dtype=t.uint8, device=self.dev)
        src[offset:offset + raw.nbytes] = t.from_numpy(raw.view(np.uint8).reshape(-1).copy()).to(self.dev)
        out = None
        pb = core._lib.load().payload_bytes(N, C, bits if bits in core.DEPTHS else 16)
        if offset or pad_stride:
            stride = (pb if offset else _align16(pb)) + pad_stride
            flat = t.zeros(F * stride + offset + 64, dtype=t.uint8, device=self.dev)
            out = flat[offset:offset + F * stride].view(F, stride) if F else flat[:0].view(0, stride)
        enc = core.analogue_batch(profile, src[offset:], fmt, F, N, C, bits, le, frame_stride=frame_stride,
                                  raw_be_ints=raw_be, check_overflow=False, out=out)
        t.cuda.synchronize()
        return enc.payload[:, :enc.nbytes].cpu().numpy(), enc.absmax.cpu().numpy()

    def digital(self, profile, payload, F, N, C, bits, le, offset=0):
        from frad_python_amd import core
        t = self.torch
        pb = payload.shape[1] if F else 0
        stride = _align16(pb) if not offset else pb
        flat = t.zeros(F * stride + offset + 64, dtype=t.uint8, device=self.dev)
        view = flat[offset:offset + F * stride].view(F, stride) if F else flat[:0].view(0, max(stride, 1))
        if F:
            view[:, :pb] = t.from_numpy(np.ascontiguousarray(payload)).to(self.dev)
        out = core.digital_batch(profile, view, F, N, C, bits, le, payload_stride=stride)
        t.cuda.synchronize()
        return out.cpu().numpy()

    def clips(self, raw: np.ndarray, fmt, N, bits, first=0, fpc=None):
        from frad_python_amd import core
        t = self.torch
        n_clips, clip_len, C = raw.shape
        dev_raw = t.from_numpy(raw.copy()).to(self.dev)
        enc = core.analogue_clips(dev_raw, fmt, N, bits, first=first, frames_per_clip=fpc)
        out = t.full((n_clips, clip_len, C), -7.0, dtype=t.float64, device=self.dev)
        core.digital_clips(enc.payload, out, N, bits, first=first, frames_per_clip=fpc)
        t.cuda.synchronize()
        return enc.payload[:, :enc.nbytes].cpu().numpy(), enc.absmax.cpu().numpy(), out.cpu().numpy()

    def crc32_frames(self, rows: np.ndarray, nbytes, offset=0):
        from frad_python_amd import core
        t = self.torch
        F, stride = rows.shape
        flat = t.zeros(F * stride + offset + 64, dtype=t.uint8, device=self.dev)
        flat[offset:offset + F * stride] = t.from_numpy(np.ascontiguousarray(rows).reshape(-1)).to(self.dev)
        view = flat[offset:offset + F * stride].view(F, stride)
        out = core.crc32_frames(view, nbytes)
        t.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32)

    def overflow_scan(self, absmax, bits, flag0=0):
        from frad_python_amd import core
        t = self.torch
        am = t.from_numpy(np.ascontiguousarray(absmax, np.float64)).to(self.dev)
        flag = t.full((), flag0, dtype=t.int32, device=self.dev)
        core.overflow_scan(am, bits, flag)
        t.cuda.synchronize()
        return int(flag.item())

    # ---- profile 1 ----
    def p1_analogue(self, raw, fmt, F, N, C, bits, srate, loss, frame_stride=None, n_valid=None):
        from frad_python_amd import core
        t = self.torch
        src = t.zeros(raw.nbytes + 64, dtype=t.uint8, device=self.dev)
        src[:raw.nbytes] = t.from_numpy(raw.view(np.uint8).reshape(-1).copy()).to(self.dev)
        q, tq = core.p1_analogue_batch(src, fmt, F, N, C, bits, srate, loss, frame_stride=frame_stride, n_valid=n_valid)
        t.cuda.synchronize()
        return q.cpu().numpy(), tq.cpu().numpy()

    def p1_digital(self, q, tq, N, C, bits, srate):
        from frad_python_amd import core
        t = self.torch
        out = core.p1_digital_batch(t.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.dev),
                                    t.from_numpy(np.ascontiguousarray(tq, np.int32)).to(self.dev), N, C, bits, srate)
        t.cuda.synchronize()
        return out.cpu().numpy()

    def p1_digital_pcm(self, q, tq, N, C, bits, srate, fmt):
        from frad_python_amd.backend.pcmformat import pcm_dtype_code, ff_format_to_numpy_type, itemsize_of
        from frad_python_amd import _lib
        t = self.torch
        F = q.shape[0]
        dt = ff_format_to_numpy_type(fmt)
        qq = t.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.dev); tt = t.from_numpy(np.ascontiguousarray(tq, np.int32)).to(self.dev)
        out = t.zeros(F * N * C * dt.itemsize + 16, dtype=t.uint8, device=self.dev)
        _lib.load().p1_digital_pcm(qq.data_ptr(), tt.data_ptr(), F, N, C, bits, srate, pcm_dtype_code(fmt), out.data_ptr(), int(t.cuda.current_stream().cuda_stream))
        t.cuda.synchronize()
        return np.frombuffer(out[:F * N * C * dt.itemsize].cpu().numpy().tobytes(), dt).reshape(F, N, C)

    def from_f64(self, x, fmt):
        from frad_python_amd import core
        from frad_python_amd.backend.pcmformat import ff_format_to_numpy_type
        t = self.torch
        out = core.from_f64(t.from_numpy(np.ascontiguousarray(x, np.float64)).to(self.dev), fmt)
        t.cuda.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), ff_format_to_numpy_type(fmt))

    def digital_pcm(self, profile, payload, F, N, C, bits, le, fmt):
        from frad_python_amd import core
        from frad_python_amd.backend.pcmformat import ff_format_to_numpy_type
        t = self.torch
        pb = payload.shape[1]
        stride = _align16(pb)
        flat = t.zeros((F, stride), dtype=t.uint8, device=self.dev)
        flat[:, :pb] = t.from_numpy(np.ascontiguousarray(payload)).to(self.dev)
        out = core.digital_batch(profile, flat, F, N, C, bits, le, payload_stride=stride, out_format=fmt)
        t.cuda.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), ff_format_to_numpy_type(fmt)).reshape(F, N, C)

    def golomb_encode(self, q, tq, offset=1):
        from frad_python_amd import core
        t = self.torch
        flat, offsets = core.p1_golomb_encode_batch(t.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.dev),
                                                    t.from_numpy(np.ascontiguousarray(tq, np.int32)).to(self.dev))
        t.cuda.synchronize()
        host, off = flat.cpu().numpy().tobytes(), offsets.cpu().numpy()
        return [host[off[i]:off[i + 1]] for i in range(q.shape[0])]

    def golomb_decode(self, bodies, N, C):
        from frad_python_amd import core
        t = self.torch
        offs = np.zeros(len(bodies) + 1, np.int64); np.cumsum([len(b) for b in bodies], out=offs[1:])
        flat = t.from_numpy(np.frombuffer(b"".join(bodies) + b"\0" * 8, np.uint8).copy()).to(self.dev)
        q, tq, st = core.p1_golomb_decode_batch(flat, t.from_numpy(offs).to(self.dev), N, C)
        t.cuda.synchronize()
        return q.cpu().numpy(), tq.cpu().numpy(), st.cpu().numpy()

    def golomb_decode_fast_only(self, bodies, N, C, with_maps):
        import ctypes as ct
        from frad_python_amd import _lib
        t = self.torch
        F = len(bodies)
        offs = np.zeros(F + 1, np.int64); np.cumsum([len(b) for b in bodies], out=offs[1:])
        flat = t.from_numpy(np.frombuffer(b"".join(bodies) + b"\0" * 8, np.uint8).copy()).to(self.dev)
        od = t.from_numpy(offs).to(self.dev)
        q = t.zeros((F, N, C), dtype=t.int32, device=self.dev); tq = t.zeros((F, 27, C), dtype=t.int32, device=self.dev)
        todo = t.full((F,), -1, dtype=t.int32, device=self.dev)
        rc = _lib.load().dll.frad_debug_golomb_decode_wave(ct.c_void_p(flat.data_ptr()), ct.c_void_p(od.data_ptr()), ct.c_int64(F), ct.c_int32(N), ct.c_int32(C),
                                                           ct.c_void_p(q.data_ptr()), ct.c_void_p(tq.data_ptr()), ct.c_void_p(todo.data_ptr()), ct.c_int32(with_maps),
                                                           ct.c_void_p(int(t.cuda.current_stream().cuda_stream)))
        assert rc == 0
        t.cuda.synchronize()
        return q.cpu().numpy(), tq.cpu().numpy(), todo.cpu().numpy()

    def p1_ola(self, frames, ratio, prev_tail=None):
        from frad_python_amd import core
        t = self.torch
        pt = t.from_numpy(np.ascontiguousarray(prev_tail)).to(self.dev) if prev_tail is not None else None
        out, nxt = core.p1_overlap_add(t.from_numpy(np.ascontiguousarray(frames)).to(self.dev), ratio, pt)
        t.cuda.synchronize()
        return out.cpu().numpy(), nxt.cpu().numpy()

    def p1_ola_pcm(self, frames, ratio, fmt, prev_tail=None):
        from frad_python_amd import core
        from frad_python_amd.backend.pcmformat import ff_format_to_numpy_type
        t = self.torch
        F, N, C = frames.shape
        pt = t.from_numpy(np.ascontiguousarray(prev_tail)).to(self.dev) if prev_tail is not None else None
        out, nxt = core.p1_overlap_add(t.from_numpy(np.ascontiguousarray(frames)).to(self.dev), ratio, pt, out_format=fmt)
        t.cuda.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), ff_format_to_numpy_type(fmt)).reshape(F, -1, C), nxt.cpu().numpy()


def oracle_frames(fo, profile, raw, fmt, F, N, C, bits, le, frame_stride=None, raw_be=True):
    """Per-frame oracle results: list of (payload bytes, depth_idx, decoded f64, absmax)."""
    dt = fo.pcm_dtype(fmt)
    hop = N if frame_stride is None else frame_stride
    flat = raw.reshape(-1, C)
    res = []
    for f in range(F):
        frame = fo.to_f64(flat[f * hop:f * hop + N], dt, be_int_quirk=raw_be)
        if profile == 0:
            X = fo.dct_channels(frame)
            am = np.max(np.abs(X)) if X.size else 0.0
            frad = fo.pack_floats(X.T.ravel(), bits, le)          # depth held fixed (escalation tested apart)
            dec = fo.p0_digital(frad, fo.DEPTHS.index(bits), C, le)
        else:
            am = np.max(np.abs(frame)) if frame.size else 0.0
            frad = fo.pack_floats(np.asarray(frame).ravel(), bits, le)
            dec = fo.p4_digital(frad, fo.DEPTHS.index(bits), C, le)
        res.append((np.frombuffer(frad, np.uint8), dec, float(am)))
    return res


def word_mismatches(a: np.ndarray, b: np.ndarray, bits: int) -> int:
    """Number of stored values (not bytes) that differ between two payloads of one frame."""
    if bits == 12:
        return int(np.count_nonzero(unpack12(a) != unpack12(b)))
    nb = bits // 8
    n = min(a.size, b.size) // nb
    return int(np.count_nonzero((a[:n * nb].reshape(n, nb) != b[:n * nb].reshape(n, nb)).any(axis=1)))


def unpack12(buf: np.ndarray) -> np.ndarray:
    nib = np.empty(buf.size * 2, np.uint16)
    nib[0::2], nib[1::2] = buf >> 4, buf & 15
    n = nib.size // 3
    t = nib[:n * 3].reshape(n, 3)
    return (t[:, 0] << 8) | (t[:, 1] << 4) | t[:, 2]


def payload_values(fo, buf: np.ndarray, bits: int, le: bool) -> np.ndarray:
    return fo.unpack_floats(buf.tobytes(), bits, le)


class OracleBridge:
    """TEST-ONLY stand-in for frad_python_amd.bridge.HipBridge: the oracle does the arithmetic, so that
    the CPU suite can check the Encoder / Decoder host logic (frame cut, overlap carry, ASFH, CRC, Golomb)
    byte-for-byte against the reference-generated streams.  Never importable from the product package."""

    def __init__(self):
        from oracle import frad_oracle as fo
        self.fo = fo

    def lossless_encode(self, profile, pcm, fmt, n_frames, N, C, bits, little_endian, raw_be_ints=True):
        fo = self.fo
        dt = fo.pcm_dtype(fmt)
        x = np.frombuffer(pcm, dt, n_frames * N * C).reshape(n_frames, N, C)
        out = []
        for f in range(n_frames):
            frame = fo.to_f64(x[f], dt, be_int_quirk=raw_be_ints)
            frad, idx, ch, sr = (fo.p4_analogue if profile == 4 else fo.p0_analogue)(frame, bits, 0, little_endian)
            out.append((frad, fo.DEPTHS[idx]))
        return out

    def lossless_decode(self, profile, payloads, N, C, bits, little_endian):
        fo = self.fo
        dig = fo.p4_digital if profile == 4 else fo.p0_digital
        return np.stack([dig(p, fo.DEPTHS.index(bits), C, little_endian) for p in payloads])

    def p1_encode(self, pcm, fmt, n_frames, N, C, bits, srate, loss_level, hop, n_valid, raw_be_ints=True):
        fo = self.fo
        dt = fo.pcm_dtype(fmt)
        x = np.frombuffer(pcm, dt).reshape(-1, C)
        q = np.zeros((n_frames, N, C), np.int32); tq = np.zeros((n_frames, 27, C), np.int32)
        for f in range(n_frames):
            a, b, aux = fo.p1_analogue_pre(fo.to_f64(x[f * hop:f * hop + n_valid], dt, be_int_quirk=raw_be_ints), bits, srate, loss_level)
            q[f] = a.reshape(N, C); tq[f] = b.reshape(27, C)
        return q, tq

    def p1_encode_bodies(self, pcm, fmt, n_frames, N, C, bits, srate, loss_level, hop, n_valid, raw_be_ints=True):
        """pre-deflate frame bodies (profile1.py:43-45) with the oracle's Exp-Golomb-Rice coder"""
        import struct
        fo = self.fo
        q, tq = self.p1_encode(pcm, fmt, n_frames, N, C, bits, srate, loss_level, hop, n_valid, raw_be_ints)
        out = []
        for f in range(n_frames):
            tg, fg = fo.golomb_encode(tq[f].reshape(-1)), fo.golomb_encode(q[f].reshape(-1))
            out.append(struct.pack(">I", len(tg)) + tg + fg)
        return out

    def p1_decode_bodies(self, bodies, N, C, bits, srate):
        """inflated frame bodies -> PCM with the oracle's decoder (profile1.py:59-77; values beyond int32 saturate as on the device)"""
        import struct
        fo = self.fo
        qs = np.zeros((len(bodies), N * C), np.int32); ts = np.zeros((len(bodies), 27 * C), np.int32)
        lim = np.iinfo(np.int32)
        for i, raw in enumerate(bodies):
            if len(raw) < 4:
                continue
            tl = struct.unpack(">I", raw[:4])[0]
            t = fo.golomb_decode(raw[4:4 + tl])[:27 * C]; q = fo.golomb_decode(raw[4 + tl:])[:N * C]
            ts[i, :t.size], qs[i, :q.size] = np.clip(t, lim.min, lim.max), np.clip(q, lim.min, lim.max)
        return self.p1_decode(qs.reshape(-1, N, C), ts.reshape(-1, 27, C), N, C, bits, srate)

    def p1_decode(self, q, tq, N, C, bits, srate):
        fo = self.fo
        return np.stack([fo.p1_digital_post(q[i].reshape(-1), tq[i].reshape(-1), fo.P1_DEPTHS.index(bits), C, srate, N)
                         for i in range(len(q))])

    def overlap_add(self, frames, ratio, prev_tail):
        fo = self.fo
        ola = fo.OverlapAdd()
        if prev_tail is not None:
            ola.fragment = np.array(prev_tail)
        out = [ola.push(f.copy(), True, ratio) for f in frames]
        return np.stack(out), ola.fragment
