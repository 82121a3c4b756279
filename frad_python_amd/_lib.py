"""ctypes binding of the C-ABI in include/frad_hip.h (libfrad_hip.so).

PyTorch is plumbing here (device memory + streams): the signatures carry raw device pointers
and sizes only.  There is no CPU fallback: if the HIP library has not been built, or no MI355X
is visible, every operator raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfrad_hip.so")

FRAD_LITTLE_ENDIAN = 1
FRAD_RAW_BE_INTS = 2

# every symbol include/frad_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "frad_abi_version": (c_int, []),
    "frad_strerror": (c_char_p, [c_int]),
    "frad_last_hip_error": (c_int, []),
    "frad_payload_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "frad_has_fast_path": (c_int, [c_int32, c_int32, c_int32]),
    "frad_plan_prepare": (c_int, [c_int32, c_int32]),
    "frad_plan_clear": (None, []),
    "frad_p0_analogue": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int64, c_int32, c_uint32,
                                 c_void_p, c_int64, c_void_p, c_void_p]),
    "frad_p0_analogue_checked": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int64, c_int32, c_uint32,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "frad_p0_overflow_scan": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "frad_p0_digital": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_uint32, c_void_p, c_void_p]),
    "frad_p0_analogue_clips": (c_int, [c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_uint32,
                                       c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "frad_p0_digital_clips": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_uint32, c_void_p, c_int64, c_void_p]),
    "frad_p4_analogue": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int64, c_int32, c_uint32,
                                 c_void_p, c_int64, c_void_p, c_void_p]),
    "frad_p4_digital": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_uint32, c_void_p, c_void_p]),
    "frad_p1_analogue": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32,
                                 c_double, c_uint32, c_void_p, c_void_p, c_void_p]),
    "frad_p1_digital": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "frad_crc32_frames": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "frad_p1_overlap_add": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "frad_p1_overlap_add_pcm": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32, c_uint32, c_void_p, c_void_p, c_void_p]),
    "frad_p1_golomb_bound": (c_size_t, [c_int32, c_int32]),
    "frad_p1_golomb_encode": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "frad_rows_compact": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "frad_p1_golomb_decode": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "frad_from_f64": (c_int, [c_void_p, c_int64, c_int32, c_uint32, c_void_p, c_void_p]),
    "frad_p0_digital_pcm": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_uint32, c_int32, c_void_p, c_void_p]),
    "frad_p4_digital_pcm": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_uint32, c_int32, c_void_p, c_void_p]),
    "frad_p1_digital_pcm": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32, c_void_p, c_void_p]),
    "frad_asfh_scan": (c_int64, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "frad_bench_copy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
}


# mirror of `frad_frame_info` (include/frad_hip.h)
def _frame_info_dtype():
    import numpy as np
    return np.dtype([("header_off", "<i8"), ("payload_off", "<i8"), ("payload_bytes", "<i8"), ("profile", "<i4"), ("ecc", "<i4"),
                     ("little_endian", "<i4"), ("depth_idx", "<i4"), ("channels", "<i4"), ("srate", "<i4"), ("fsize", "<i4"),
                     ("overlap_ratio", "<i4"), ("ecc_dsize", "<i4"), ("ecc_codesize", "<i4"), ("force_flush", "<i4"), ("crc", "<u4")])


FRAME_INFO_DTYPE = _frame_info_dtype()
assert FRAME_INFO_DTYPE.itemsize == 72


class FradError(RuntimeError):
    def __init__(self, status: int, message: str, hip_error: int = 0):
        super().__init__(f"libfrad_hip: {message} (status {status}" + (f", hipError {hip_error})" if hip_error else ")"))
        self.status = status
        self.hip_error = hip_error


class FradLib:
    """One loaded copy of the C-ABI library; methods take raw pointers (ints) and sizes."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP transform core has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        self.path = path
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and must be the one that gets loaded.  If this
        # library came first it would bind /opt/rocm's copy, torch would then bring a second runtime, and every launch here
        # would fail with hipErrorNoDevice (seen when build() and smoke() run in one process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self.dll, name)           # AttributeError if the library lacks a symbol
            fn.restype, fn.argtypes = res, args

    def _check(self, rc: int):
        if rc != 0:
            raise FradError(rc, self.dll.frad_strerror(rc).decode(), self.dll.frad_last_hip_error())

    def payload_bytes(self, N, C, bits):
        return int(self.dll.frad_payload_bytes(N, C, bits))

    def has_fast_path(self, N, C, f32=False):
        return bool(self.dll.frad_has_fast_path(N, C, int(f32)))

    def plan_prepare(self, N, f32=False):
        self._check(self.dll.frad_plan_prepare(N, int(f32)))

    def p0_analogue(self, pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, stream=0):
        self._check(self.dll.frad_p0_analogue(pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload,
                                               payload_stride, absmax, stream))

    def p0_analogue_checked(self, pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, flag, stream=0):
        self._check(self.dll.frad_p0_analogue_checked(pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload,
                                                       payload_stride, absmax, flag, stream))

    def crc32_frames(self, data, stride, n_frames, nbytes, out, stream=0):
        self._check(self.dll.frad_crc32_frames(data, stride, n_frames, nbytes, out, stream))

    def p0_overflow_scan(self, absmax, n_frames, bits, flag, stream=0):
        self._check(self.dll.frad_p0_overflow_scan(absmax, n_frames, bits, flag, stream))

    def p0_digital(self, payload, payload_stride, n_frames, N, C, bits, flags, out, stream=0):
        self._check(self.dll.frad_p0_digital(payload, payload_stride, n_frames, N, C, bits, flags, out, stream))

    def p0_analogue_clips(self, pcm, dtype, n_clips, clip_stride, frames_per_clip, N, C, bits, flags, payload, payload_stride, absmax, flag, stream=0):
        self._check(self.dll.frad_p0_analogue_clips(pcm, dtype, n_clips, clip_stride, frames_per_clip, N, C, bits, flags, payload,
                                                     payload_stride, absmax, flag, stream))

    def p0_digital_clips(self, payload, payload_stride, n_clips, frames_per_clip, N, C, bits, flags, out, out_clip_stride, stream=0):
        self._check(self.dll.frad_p0_digital_clips(payload, payload_stride, n_clips, frames_per_clip, N, C, bits, flags, out, out_clip_stride, stream))

    def p4_analogue(self, pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, stream=0):
        self._check(self.dll.frad_p4_analogue(pcm, dtype, n_frames, N, C, frame_stride, bits, flags, payload,
                                               payload_stride, absmax, stream))

    def p4_digital(self, payload, payload_stride, n_frames, N, C, bits, flags, out, stream=0):
        self._check(self.dll.frad_p4_digital(payload, payload_stride, n_frames, N, C, bits, flags, out, stream))

    def p1_analogue(self, pcm, dtype, n_frames, N, C, frame_stride, n_valid, bits, srate, loss_level, flags, q, tq, stream=0):
        self._check(self.dll.frad_p1_analogue(pcm, dtype, n_frames, N, C, frame_stride, n_valid, bits, srate,
                                               loss_level, flags, q, tq, stream))

    def p1_digital(self, q, tq, n_frames, N, C, bits, srate, out, stream=0):
        self._check(self.dll.frad_p1_digital(q, tq, n_frames, N, C, bits, srate, out, stream))

    def p1_golomb_bound(self, N, C):
        return int(self.dll.frad_p1_golomb_bound(N, C))

    def p1_golomb_encode(self, q, tq, n_frames, N, C, bodies, body_stride, body_bytes, stream=0):
        self._check(self.dll.frad_p1_golomb_encode(q, tq, n_frames, N, C, bodies, body_stride, body_bytes, stream))

    def rows_compact(self, rows, row_stride, row_bytes, n_rows, out, offsets, stream=0):
        self._check(self.dll.frad_rows_compact(rows, row_stride, row_bytes, n_rows, out, offsets, stream))

    def p1_golomb_decode(self, bodies, offsets, n_frames, N, C, q, tq, status, stream=0):
        self._check(self.dll.frad_p1_golomb_decode(bodies, offsets, n_frames, N, C, q, tq, status, stream))

    def from_f64(self, pcm, n_values, out_dtype, out, stream=0, flags=FRAD_RAW_BE_INTS):
        self._check(self.dll.frad_from_f64(pcm, n_values, out_dtype, flags, out, stream))

    def p0_digital_pcm(self, payload, payload_stride, n_frames, N, C, bits, flags, out_dtype, out, stream=0):
        self._check(self.dll.frad_p0_digital_pcm(payload, payload_stride, n_frames, N, C, bits, flags, out_dtype, out, stream))

    def p4_digital_pcm(self, payload, payload_stride, n_frames, N, C, bits, flags, out_dtype, out, stream=0):
        self._check(self.dll.frad_p4_digital_pcm(payload, payload_stride, n_frames, N, C, bits, flags, out_dtype, out, stream))

    def p1_digital_pcm(self, q, tq, n_frames, N, C, bits, srate, out_dtype, out, stream=0, flags=FRAD_RAW_BE_INTS):
        self._check(self.dll.frad_p1_digital_pcm(q, tq, n_frames, N, C, bits, srate, out_dtype, flags, out, stream))

    def asfh_scan(self, data, start: int = 0, max_frames: int = 0):
        """frad_asfh_scan over a bytes-like object -> (numpy structured table, next_pos, stop_reason)"""
        import numpy as np
        view = memoryview(data)
        n = view.nbytes
        cap = max_frames or max(16, (n - start) // 9 + 1)
        cap = min(cap, 1 << 22)
        table = np.zeros(cap, FRAME_INFO_DTYPE)
        nxt, why = ctypes.c_int64(0), ctypes.c_int32(0)
        buf = (ctypes.c_char * n).from_buffer_copy(view) if view.readonly and not isinstance(data, (bytes, bytearray)) else None
        ptr = ctypes.cast(ctypes.c_char_p(data), c_void_p) if isinstance(data, bytes) else \
            ctypes.addressof((ctypes.c_char * n).from_buffer(data)) if buf is None else ctypes.addressof(buf)
        rows = self.dll.frad_asfh_scan(ptr, n, start, table.ctypes.data, cap, ctypes.byref(nxt), ctypes.byref(why))
        if rows < 0:
            self._check(int(rows))
        return table[:rows], int(nxt.value), int(why.value)

    def bench_copy(self, src, dst, nbytes, stream=0):
        self._check(self.dll.frad_bench_copy(src, dst, nbytes, stream))

    def p1_overlap_add_pcm(self, frames, n_frames, N, C, ratio, prev_tail, out_dtype, out, next_tail, stream=0, flags=FRAD_RAW_BE_INTS):
        self._check(self.dll.frad_p1_overlap_add_pcm(frames, n_frames, N, C, ratio, prev_tail, out_dtype, flags, out, next_tail, stream))

    def p1_overlap_add(self, frames, n_frames, N, C, ratio, prev_tail, out, next_tail, stream=0):
        self._check(self.dll.frad_p1_overlap_add(frames, n_frames, N, C, ratio, prev_tail, out, next_tail, stream))


_lib: FradLib | None = None


def load() -> FradLib:
    """The product library (HIP, gfx950).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = FradLib(LIB_PATH)
    return _lib
