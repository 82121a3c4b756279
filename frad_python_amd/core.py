"""Batched operators of the transform core on PyTorch-ROCm tensors.

One call = one HIP launch over ``n_frames`` independent frames (the reference runs one frame
per call through ``fourier.profileN.analogue/digital``, src/libfrad/encoder.py:96-100 and
decoder.py:70-74).  Tensors only carry device memory and the current stream into the C-ABI
(include/frad_hip.h); all arithmetic happens in libfrad_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .backend.pcmformat import itemsize_of, pcm_dtype_code

DEPTHS = (12, 16, 24, 32, 48, 64)                      # ref: fourier/profile0.py:4, profile4.py:4
# largest finite value of each depth's storage float (ref: profile0.py:6-13 FLOAT_DR)
FLOAT_MAX = {12: 65504.0, 16: 65504.0, 24: float(np.finfo("f4").max), 32: float(np.finfo("f4").max),
             48: float(np.finfo("f8").max), 64: float(np.finfo("f8").max)}
_ESCALATE = {12: 16, 16: 24, 24: 32, 32: 48, 48: 64, 64: 128}


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live in MI355X device memory (got a {t.device} tensor); "
                           "the transform core has no CPU path")
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")


def _require_cuda_any(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live in MI355X device memory (got a {t.device} tensor); "
                           "the transform core has no CPU path")


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _align16(n: int) -> int:
    return (n + 15) // 16 * 16


def escalate_depth(absmax: float, bits: int) -> int:
    """The reference's overflow loop (profile0.py:24-26): NaN never escalates, +Inf raises."""
    while absmax > FLOAT_MAX[bits]:
        bits = _ESCALATE[bits]
        if bits == 128:
            raise OverflowError("Overflow with reaching the max bit depth.")
    return bits


@dataclass
class EncodedBatch:
    """Payloads of one batched ``analogue`` call.

    ``payload[i, :nbytes]`` is frame i at the requested depth.  A frame whose transform exceeds the
    storage float's range (profile0.py:24-26) is listed in ``escalated`` as
    ``{frame: (payload_row_tensor, bits)}`` at the deeper format the reference would pick."""
    payload: torch.Tensor            # uint8 [n_frames, stride]
    nbytes: int
    bits: int
    absmax: torch.Tensor             # float64 [n_frames]
    escalated: dict

    def frame_bytes(self, i: int) -> tuple[bytes, int]:
        if i in self.escalated:
            row, bits = self.escalated[i]
            return bytes(row.cpu().numpy()), bits
        return bytes(self.payload[i, :self.nbytes].cpu().numpy()), self.bits


def analogue_batch(profile: int, pcm: torch.Tensor, pcm_format: str, n_frames: int, N: int, C: int, bits: int,
                   little_endian: bool = False, *, frame_stride: int | None = None, raw_be_ints: bool = True,
                   check_overflow: bool = True, out: torch.Tensor | None = None,
                   absmax: torch.Tensor | None = None, overflow_flag: torch.Tensor | None = None) -> EncodedBatch:
    """Profile 0 (DCT) or 4 (PCM) ``analogue`` over a batch of frames.

    ``pcm`` is the raw interleaved PCM (any tensor dtype; ``pcm_format`` names the element type as
    the reference's CLI does, e.g. ``s16le``), frame i starting ``frame_stride`` (default N)
    sample-frames after frame i-1.  ``overflow_flag`` (profile 0, device int32 scalar): the batch form of the reference's
    overflow test in the same pass -- set to 1 when a frame needs a deeper format, read it when the answer is needed."""
    _require_cuda(pcm, "pcm")
    if bits not in DEPTHS:
        bits = 16                                         # ref: profile0.py:15
    lib = _lib.load()
    code = pcm_dtype_code(pcm_format)
    stride_frames = N if frame_stride is None else frame_stride
    need = ((n_frames - 1) * stride_frames + N) * C * itemsize_of(code) if n_frames else 0
    if pcm.numel() * pcm.element_size() < need:
        raise ValueError(f"pcm holds {pcm.numel() * pcm.element_size()} bytes, {need} needed")
    nbytes = lib.payload_bytes(N, C, bits)
    stride = _align16(nbytes)
    if out is None:
        out = torch.empty((n_frames, stride), dtype=torch.uint8, device=pcm.device)
    if absmax is None:
        absmax = torch.empty(n_frames, dtype=torch.float64, device=pcm.device)
    flags = (int(little_endian) * _lib.FRAD_LITTLE_ENDIAN) | (int(raw_be_ints) * _lib.FRAD_RAW_BE_INTS)
    fn = lib.p4_analogue if profile == 4 else lib.p0_analogue
    with torch.cuda.device(pcm.device):
        if overflow_flag is not None and profile != 4:
            _require_cuda(overflow_flag, "overflow_flag")
            if overflow_flag.dtype != torch.int32:
                raise TypeError("overflow_flag must be int32")
            lib.p0_analogue_checked(pcm.data_ptr(), code, n_frames, N, C, stride_frames, bits, flags, out.data_ptr(), out.stride(0),
                                    absmax.data_ptr(), overflow_flag.data_ptr(), _stream_ptr())
        else:
            fn(pcm.data_ptr(), code, n_frames, N, C, stride_frames, bits, flags, out.data_ptr(), out.stride(0),
               absmax.data_ptr(), _stream_ptr())
    escalated = {}
    if check_overflow and n_frames:
        over = absmax > FLOAT_MAX[bits]                   # NaN compares False, as in the reference
        idx = over.nonzero().flatten()                    # (one host sync: the number of offenders sizes the re-dispatch)
        if idx.numel():
            # the offenders again, one launch per deeper format the reference's loop would settle on (profile0.py:24-26)
            isz = itemsize_of(code)
            flat = pcm.reshape(-1).view(torch.uint8)
            frame_bytes_n = N * C * isz
            byte_idx = torch.arange(frame_bytes_n, device=pcm.device)
            am_host = absmax[idx].cpu().tolist()
            by_depth: dict[int, list[int]] = {}
            for i, a in zip(idx.tolist(), am_host):
                by_depth.setdefault(escalate_depth(float(a), bits), []).append(i)
            for deeper, frames in by_depth.items():
                starts = torch.tensor(frames, device=pcm.device, dtype=torch.int64) * (stride_frames * C * isz)
                gathered = flat[(starts[:, None] + byte_idx[None, :]).reshape(-1)]     # [len(frames), N*C*isz] contiguous
                sub = analogue_batch(profile, gathered, pcm_format, len(frames), N, C, deeper, little_endian,
                                     raw_be_ints=raw_be_ints, check_overflow=False)
                for j, i in enumerate(frames):
                    escalated[i] = (sub.payload[j, :sub.nbytes], deeper)
    return EncodedBatch(out, nbytes, bits, absmax, escalated)


def analogue_clips(clips: torch.Tensor, pcm_format: str, N: int, bits: int, little_endian: bool = False, *,
                   first: int = 0, frames_per_clip: int | None = None, raw_be_ints: bool = True,
                   out: torch.Tensor | None = None, absmax: torch.Tensor | None = None,
                   overflow_flag: torch.Tensor | None = None) -> EncodedBatch:
    """Profile 0 ``analogue`` over a resident batch of equally long clips ``[n_clips, clip_len, C]``, consumed in place
    (frad_p0_analogue_clips; the reference cuts every clip into frames on its own, encoder.py:72-93): ``frames_per_clip``
    frames of N sample-frames per clip, the first one ``first`` sample-frames into the clip (default: as many whole
    frames as fit behind ``first``).  A clip's shorter last frame is a second call with its own N and ``first``.
    Payload row ``c * frames_per_clip + i`` is frame i of clip c.  The overflow test is the device flag only."""
    _require_cuda(clips, "clips")
    if clips.dim() != 3:
        raise ValueError("clips must be [n_clips, clip_len, C]")
    if bits not in DEPTHS:
        bits = 16
    lib = _lib.load()
    code = pcm_dtype_code(pcm_format)
    n_clips, clip_len, C = clips.shape
    if clips.element_size() != itemsize_of(code):
        raise ValueError("clips' element size does not match pcm_format")
    fpc = (clip_len - first) // N if frames_per_clip is None else frames_per_clip
    if fpc < 1 or first < 0 or first + fpc * N > clip_len:
        raise ValueError("the frames do not fit the clip")
    n_frames = n_clips * fpc
    nbytes = lib.payload_bytes(N, C, bits)
    if out is None:
        out = torch.empty((n_frames, _align16(nbytes)), dtype=torch.uint8, device=clips.device)
    if absmax is None:
        absmax = torch.empty(n_frames, dtype=torch.float64, device=clips.device)
    flags = (int(little_endian) * _lib.FRAD_LITTLE_ENDIAN) | (int(raw_be_ints) * _lib.FRAD_RAW_BE_INTS)
    with torch.cuda.device(clips.device):
        lib.p0_analogue_clips(clips.data_ptr() + first * C * clips.element_size(), code, n_clips, clip_len, fpc, N, C, bits, flags,
                              out.data_ptr(), out.stride(0), absmax.data_ptr(),
                              overflow_flag.data_ptr() if overflow_flag is not None else 0, _stream_ptr())
    return EncodedBatch(out, nbytes, bits, absmax, {})


def digital_clips(payload: torch.Tensor, out: torch.Tensor, N: int, bits: int, little_endian: bool = False, *,
                  first: int = 0, frames_per_clip: int | None = None) -> torch.Tensor:
    """Profile 0 ``digital`` of a clip batch straight into ``out`` = float64 ``[n_clips, clip_len, C]`` (frad_p0_digital_clips):
    payload row ``c * frames_per_clip + i`` lands at ``out[c, first + i*N : first + (i+1)*N]``."""
    _require_cuda(payload, "payload"); _require_cuda(out, "out")
    if out.dim() != 3 or out.dtype != torch.float64:
        raise ValueError("out must be float64 [n_clips, clip_len, C]")
    lib = _lib.load()
    n_clips, clip_len, C = out.shape
    fpc = (clip_len - first) // N if frames_per_clip is None else frames_per_clip
    if fpc < 1 or first < 0 or first + fpc * N > clip_len or payload.shape[0] < n_clips * fpc:
        raise ValueError("the frames do not fit the clip / the payload batch")
    flags = int(little_endian) * _lib.FRAD_LITTLE_ENDIAN
    with torch.cuda.device(out.device):
        lib.p0_digital_clips(payload.data_ptr(), payload.stride(0), n_clips, fpc, N, C, bits, flags,
                             out.data_ptr() + first * C * 8, clip_len, _stream_ptr())
    return out


def overflow_scan(absmax: torch.Tensor, bits: int, flag: torch.Tensor) -> None:
    """``flag |= any(absmax > FLOAT_MAX[bits])`` in one launch (profile0.py:24-26 over a batch).

    ``flag`` is a device int32 scalar kept across batches; read it when the answer is needed."""
    _require_cuda(absmax, "absmax"); _require_cuda(flag, "flag")
    if absmax.dtype != torch.float64 or flag.dtype != torch.int32:
        raise TypeError("absmax must be float64 and flag int32")
    with torch.cuda.device(absmax.device):
        _lib.load().p0_overflow_scan(absmax.data_ptr(), absmax.numel(), bits, flag.data_ptr(), _stream_ptr())


def crc32_frames(payload: torch.Tensor, nbytes: int) -> torch.Tensor:
    """zlib.crc32 of ``payload[i, :nbytes]`` for every row, as int32 bit patterns on the device
    (the checksum ASFH.write stores in a lossless frame header, tools/asfh.py:51-73)."""
    _require_cuda_any(payload, "payload")
    if payload.dtype != torch.uint8 or payload.dim() != 2 or payload.stride(1) != 1 or payload.shape[1] < nbytes:
        raise ValueError("payload must be a uint8 [n_frames, >= nbytes] tensor with unit column stride")
    out = torch.empty(payload.shape[0], dtype=torch.int32, device=payload.device)
    with torch.cuda.device(payload.device):
        _lib.load().crc32_frames(payload.data_ptr(), payload.stride(0), payload.shape[0], nbytes, out.data_ptr(), _stream_ptr())
    return out


def _pcm_out_tensor(fmt: str, shape, device) -> torch.Tensor:
    """uint8 storage for `shape` elements of PCM format `fmt` (torch has no big-endian or unsigned 16/32/64 dtypes)"""
    n = 1
    for d in shape:
        n *= d
    return torch.empty(n * itemsize_of(pcm_dtype_code(fmt)), dtype=torch.uint8, device=device)


def from_f64(pcm: torch.Tensor, out_format: str, *, raw_be_ints: bool = True) -> torch.Tensor:
    """``from_f64(pcm, fmt).astype(fmt)`` as the reference's caller applies it to decoded blocks (pcmformat.py:49-62,
    src/decoder.py:23): float64 tensor -> the bytes of PCM format ``out_format`` (uint8 tensor)."""
    _require_cuda(pcm, "pcm")
    if pcm.dtype != torch.float64 or not pcm.is_contiguous():
        raise TypeError("pcm must be contiguous float64")
    out = _pcm_out_tensor(out_format, pcm.shape, pcm.device)
    with torch.cuda.device(pcm.device):
        _lib.load().from_f64(pcm.data_ptr(), pcm.numel(), pcm_dtype_code(out_format), out.data_ptr(), _stream_ptr(),
                             int(raw_be_ints) * _lib.FRAD_RAW_BE_INTS)
    return out


def digital_batch(profile: int, payload: torch.Tensor, n_frames: int, N: int, C: int, bits: int,
                  little_endian: bool = False, *, payload_stride: int | None = None,
                  out: torch.Tensor | None = None, out_format: str | None = None) -> torch.Tensor:
    """Profile 0 / 4 ``digital`` over a batch: uint8 payload rows -> float64 [n_frames, N, C]; with ``out_format`` the
    samples leave the device already narrowed to that PCM format (uint8 tensor of its bytes, see ``from_f64``)."""
    _require_cuda(payload, "payload")
    lib = _lib.load()
    nbytes = lib.payload_bytes(N, C, bits)
    if payload_stride is None:
        payload_stride = payload.stride(0) if payload.dim() == 2 else nbytes
    if payload.numel() * payload.element_size() < ((n_frames - 1) * payload_stride + nbytes if n_frames else 0):
        raise ValueError("payload tensor is smaller than n_frames frames")
    flags = int(little_endian) * _lib.FRAD_LITTLE_ENDIAN
    if out_format is not None:
        flags |= _lib.FRAD_RAW_BE_INTS                          # the reference's from_f64 does not recognise big-endian ints
        if out is None:
            out = _pcm_out_tensor(out_format, (n_frames, N, C), payload.device)
        fn = lib.p4_digital_pcm if profile == 4 else lib.p0_digital_pcm
        with torch.cuda.device(payload.device):
            fn(payload.data_ptr(), payload_stride, n_frames, N, C, bits, flags, pcm_dtype_code(out_format), out.data_ptr(), _stream_ptr())
        return out
    if out is None:
        out = torch.empty((n_frames, N, C), dtype=torch.float64, device=payload.device)
    fn = lib.p4_digital if profile == 4 else lib.p0_digital
    with torch.cuda.device(payload.device):
        fn(payload.data_ptr(), payload_stride, n_frames, N, C, bits, flags, out.data_ptr(), _stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------
# profile 1 (psychoacoustic quantiser): the integer arrays either side of the host entropy coder
# ---------------------------------------------------------------------------------------------
P1_DEPTHS = (8, 12, 16, 24, 32, 48, 64)                # ref: fourier/profile1.py:7
P1_BANDS = 27


def p1_analogue_batch(pcm: torch.Tensor, pcm_format: str, n_frames: int, N: int, C: int, bits: int, srate: int,
                      loss_level: float, *, frame_stride: int | None = None, n_valid: int | None = None,
                      raw_be_ints: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """``profile1.analogue`` up to the Exp-Golomb coder (profile1.py:15-40), batched.

    Frame i reads ``n_valid`` (default N) sample-frames at ``pcm + i*frame_stride`` (the hop when the
    encoder overlaps) and is zero-padded to the compact frame size N.  Returns ``q`` int32
    [n_frames, N, C] and ``tq`` int32 [n_frames, 27, C]."""
    _require_cuda(pcm, "pcm")
    lib = _lib.load()
    code = pcm_dtype_code(pcm_format)
    hop = N if frame_stride is None else frame_stride
    nv = N if n_valid is None else n_valid
    need = ((n_frames - 1) * hop + nv) * C * itemsize_of(code) if n_frames else 0
    if pcm.numel() * pcm.element_size() < need:
        raise ValueError(f"pcm holds {pcm.numel() * pcm.element_size()} bytes, {need} needed")
    q = torch.empty((n_frames, N, C), dtype=torch.int32, device=pcm.device)
    tq = torch.empty((n_frames, P1_BANDS, C), dtype=torch.int32, device=pcm.device)
    flags = int(raw_be_ints) * _lib.FRAD_RAW_BE_INTS
    with torch.cuda.device(pcm.device):
        lib.p1_analogue(pcm.data_ptr(), code, n_frames, N, C, hop, nv, bits, srate, float(loss_level), flags,
                        q.data_ptr(), tq.data_ptr(), _stream_ptr())
    return q, tq


def p1_golomb_encode_batch(q: torch.Tensor, tq: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """The Exp-Golomb-Rice stage of ``profile1.analogue`` (profile1.py:43-45, p1tools.py:46-60) on the device.

    ``q`` int32 [n_frames, N, C], ``tq`` int32 [n_frames, 27, C] -> ``(bodies uint8 [total], offsets int64
    [n_frames + 1])``: frame i's pre-deflate body ('>I' len + Golomb(tq) + Golomb(q)) is
    ``bodies[offsets[i]:offsets[i+1]]``.  Deflate stays on the host."""
    _require_cuda(q, "q"); _require_cuda(tq, "tq")
    if q.dtype != torch.int32 or tq.dtype != torch.int32 or not q.is_contiguous() or not tq.is_contiguous():
        raise TypeError("q and tq must be contiguous int32")
    n_frames, N, C = q.shape
    lib = _lib.load()
    stride = lib.p1_golomb_bound(N, C)
    rows = torch.empty((n_frames, stride), dtype=torch.uint8, device=q.device)
    nbytes = torch.empty(n_frames, dtype=torch.int64, device=q.device)
    offsets = torch.empty(n_frames + 1, dtype=torch.int64, device=q.device)
    with torch.cuda.device(q.device):
        lib.p1_golomb_encode(q.data_ptr(), tq.data_ptr(), n_frames, N, C, rows.data_ptr(), stride, nbytes.data_ptr(), _stream_ptr())
        lib.rows_compact(rows.data_ptr(), stride, nbytes.data_ptr(), n_frames, 0, offsets.data_ptr(), _stream_ptr())
        total = int(offsets[-1].item()) if n_frames else 0      # the one host read: the size of the result
        out = torch.empty(total, dtype=torch.uint8, device=q.device)
        lib.rows_compact(rows.data_ptr(), stride, nbytes.data_ptr(), n_frames, out.data_ptr(), offsets.data_ptr(), _stream_ptr())
    return out, offsets


def p1_golomb_decode_batch(bodies: torch.Tensor, offsets: torch.Tensor, N: int, C: int):
    """The two ``exp_golomb_rice_decode`` calls + ``untrim`` of ``profile1.digital`` (profile1.py:59-64,
    p1tools.py:62-74): inflated bodies (uint8, frame i at ``offsets[i]:offsets[i+1]``) -> ``(q, tq, status)``."""
    _require_cuda(bodies, "bodies"); _require_cuda(offsets, "offsets")
    if bodies.dtype != torch.uint8 or offsets.dtype != torch.int64:
        raise TypeError("bodies must be uint8 and offsets int64")
    n_frames = offsets.numel() - 1
    q = torch.empty((n_frames, N, C), dtype=torch.int32, device=bodies.device)
    tq = torch.empty((n_frames, P1_BANDS, C), dtype=torch.int32, device=bodies.device)
    status = torch.empty(max(n_frames, 1), dtype=torch.int32, device=bodies.device)
    with torch.cuda.device(bodies.device):
        _lib.load().p1_golomb_decode(bodies.data_ptr(), offsets.data_ptr(), n_frames, N, C, q.data_ptr(), tq.data_ptr(),
                                     status.data_ptr(), _stream_ptr())
    return q, tq, status[:n_frames]


def p1_digital_batch(q: torch.Tensor, tq: torch.Tensor, N: int, C: int, bits: int, srate: int) -> torch.Tensor:
    """``profile1.digital`` from the decoded integers on (profile1.py:65-77): float64 [n_frames, N, C]."""
    _require_cuda(q, "q"); _require_cuda(tq, "tq")
    if q.dtype != torch.int32 or tq.dtype != torch.int32:
        raise TypeError("q and tq must be int32")
    n_frames = q.shape[0]
    out = torch.empty((n_frames, N, C), dtype=torch.float64, device=q.device)
    with torch.cuda.device(q.device):
        _lib.load().p1_digital(q.data_ptr(), tq.data_ptr(), n_frames, N, C, bits, srate, out.data_ptr(), _stream_ptr())
    return out


def p1_overlap_add(frames: torch.Tensor, overlap_ratio: int, prev_tail: torch.Tensor | None = None, out_format: str | None = None):
    """The decoder's Hann cross-fade over consecutive decoded frames (decoder.py:28-46).

    ``frames`` float64 [n_frames, N, C]; returns ``(out [n_frames, cut, C], next_tail [N - cut, C])``
    with cut = N*(ratio-1)//ratio; ``prev_tail`` is the previous batch's ``next_tail`` (or None).  ``out_format``: the
    caller's ``from_f64(...).astype(fmt)`` (src/decoder.py:23) applied in the same pass -- ``out`` is then a uint8 tensor
    holding ``[n_frames, cut, C]`` elements of that PCM format (frad_p1_overlap_add_pcm); the tail stays float64."""
    _require_cuda(frames, "frames")
    n_frames, N, C = frames.shape
    cut = N * (overlap_ratio - 1) // overlap_ratio
    nxt = torch.empty((N - cut, C), dtype=torch.float64, device=frames.device)
    if prev_tail is not None:
        _require_cuda(prev_tail, "prev_tail")
        if tuple(prev_tail.shape) != (N - cut, C):
            raise ValueError("prev_tail must be [N - cut, C]")
    pt = prev_tail.data_ptr() if prev_tail is not None else 0
    with torch.cuda.device(frames.device):
        if out_format is not None and out_format not in ("f64le",):
            code = pcm_dtype_code(out_format)
            out = torch.empty(n_frames * cut * C * itemsize_of(code) + 16, dtype=torch.uint8, device=frames.device)[:n_frames * cut * C * itemsize_of(code)]
            _lib.load().p1_overlap_add_pcm(frames.data_ptr(), n_frames, N, C, overlap_ratio, pt, code, out.data_ptr(), nxt.data_ptr(), _stream_ptr())
        else:
            out = torch.empty((n_frames, cut, C), dtype=torch.float64, device=frames.device)
            _lib.load().p1_overlap_add(frames.data_ptr(), n_frames, N, C, overlap_ratio, pt, out.data_ptr(), nxt.data_ptr(), _stream_ptr())
    return out, nxt
