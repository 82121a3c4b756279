"""PCM sample formats of the drop-in boundary.

Mirrors ``ff_format_to_numpy_type`` of the reference (src/libfrad/backend/pcmformat.py:4-32:
same ffmpeg-style names, same "unknown name -> message on stderr + exit(1)") and maps each
format onto the ``frad_pcm_dtype`` code of the C-ABI (include/frad_hip.h).  The int -> float
normalisation itself (reference ``to_f64``, :34-47) is not done on the host: it is fused into
the load stage of the HIP kernels.
"""
from __future__ import annotations

import sys

import numpy as np

# name -> (numpy dtype string, frad_pcm_dtype code).  Codes: kind*8 + log2(itemsize)*2 + big_endian
# with kind 0 = unsigned, 1 = signed, 2 = float (see include/frad_hip.h).
_FORMATS = {}
for _kind, _letter, _k in (("u", "u", 0), ("s", "i", 1), ("f", "f", 2)):
    for _sz in (1, 2, 4, 8):
        if _kind == "f" and _sz == 1:
            continue
        for _be in (0, 1):
            _code = _k * 8 + {1: 0, 2: 1, 4: 2, 8: 3}[_sz] * 2 + _be
            if _sz == 1:
                if _be:
                    continue
                _FORMATS[f"{_kind}8"] = (f"{_letter}1", _code)
            else:
                _FORMATS[f"{_kind}{_sz * 8}{'be' if _be else 'le'}"] = \
                    (f"{'>' if _be else '<'}{_letter}{_sz}", _code)


def ff_format_to_numpy_type(x: str) -> np.dtype:
    try:
        return np.dtype(_FORMATS[x.lower()][0])
    except KeyError:
        print(f"Invalid format: {x}", file=sys.stderr)
        sys.exit(1)


def pcm_dtype_code(fmt) -> int:
    """``frad_pcm_dtype`` code for a format name or a numpy dtype."""
    if isinstance(fmt, str):
        if fmt.lower() not in _FORMATS:
            raise ValueError(f"Invalid format: {fmt}")
        return _FORMATS[fmt.lower()][1]
    dt = np.dtype(fmt)
    kind = {"u": 0, "i": 1, "f": 2}[dt.kind]
    be = int(dt.itemsize > 1 and (dt.byteorder == ">" or (dt.byteorder == "=" and sys.byteorder == "big")))
    return kind * 8 + {1: 0, 2: 1, 4: 2, 8: 3}[dt.itemsize] * 2 + be


def is_float_format(code: int) -> bool:
    return code // 8 == 2


def itemsize_of(code: int) -> int:
    return 1 << ((code % 8) // 2)


FORMAT_NAMES = tuple(_FORMATS)


def from_f64(pcm: np.ndarray, fmt) -> np.ndarray:
    """Host form of the decoder's output conversion, ``from_f64(pcm, fmt).astype(fmt)`` as the reference's caller
    applies it (src/libfrad/backend/pcmformat.py:49-62, src/decoder.py:23): floats are cast, native-order integers are
    scaled by 2^(w-1) (unsigned: after adding 1) and truncated by ``astype``; big-endian integer formats are not
    recognised by the reference's dtype comparison, so their samples are truncated unscaled.  The device form is
    ``frad_from_f64`` (csrc/frad_epilogue.hip); this one serves the decoder's rare host-side paths."""
    dt = ff_format_to_numpy_type(fmt) if isinstance(fmt, str) else np.dtype(fmt)
    x = np.asarray(pcm, np.float64)
    if dt.kind in "iu" and dt.isnative:
        w = 8 * dt.itemsize
        x = (x + 1.0 if dt.kind == "u" else x) * float(2 ** (w - 1))
    with np.errstate(all="ignore"):
        return x.astype(dt)
