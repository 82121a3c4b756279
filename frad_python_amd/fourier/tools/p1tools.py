"""Host entropy stage of profile 1: Exp-Golomb-Rice coding of the quantiser's integers.

Format (reference src/libfrad/fourier/tools/p1tools.py:46-74): one byte k = ceil(log2(max|v|)), then per
value the zig-zag number z (v > 0: 2v-1, else -2v) as m zero bits followed by the (m+k+1)-bit binary of
z + 2^k; the bit string is zero-padded to a byte.  Vectorised with NumPy (the reference builds Python
strings bit by bit); byte-identical output is pinned by the golden vectors."""
from __future__ import annotations

import struct

import numpy as np


def exp_golomb_rice_encode(data: np.ndarray) -> bytes:
    data = np.asarray(data).astype(np.int64).ravel()
    if not data.size:
        return b"\x00"
    dmax = int(np.abs(data).max())
    k = int(np.ceil(np.log2(dmax))) if dmax else 0
    z = np.where(data > 0, 2 * data - 1, -2 * data) + (1 << k)          # < 2^40 in practice
    nbits = np.zeros(z.shape, np.int64)
    t = z.copy()
    while True:                                                         # exact bit length, no float log
        live = t > 0
        if not live.any():
            break
        nbits += live
        t >>= 1
    m = nbits - (k + 1)
    total = m + nbits
    ends = np.cumsum(total)
    starts = ends - total
    bits = np.zeros(int(ends[-1]), np.uint8)
    for j in range(int(nbits.max())):
        sel = nbits > j
        bits[starts[sel] + m[sel] + j] = (z[sel] >> (nbits[sel] - 1 - j)) & 1
    return struct.pack("B", k) + np.packbits(bits).tobytes()


def exp_golomb_rice_decode(dbytes: bytes) -> np.ndarray:
    k = dbytes[0]
    bits = np.unpackbits(np.frombuffer(dbytes, np.uint8, offset=1))
    n = bits.size
    ones = np.flatnonzero(bits)
    weights_cache = {}
    out = []
    pos = 0
    oi = 0
    nones = ones.size
    while pos < n:
        while oi < nones and ones[oi] < pos:
            oi += 1
        if oi >= nones:
            break                                                       # only padding zeros are left
        m = int(ones[oi]) - pos
        ln = 2 * m + k + 1
        word = bits[pos:pos + ln]
        w = weights_cache.get(word.size)
        if w is None:
            w = weights_cache[word.size] = (1 << np.arange(word.size - 1, -1, -1, dtype=np.int64))
        v = int(word.astype(np.int64) @ w) - (1 << k)
        out.append((v + 1) >> 1 if v & 1 else -(v >> 1))
        pos += ln
    return np.array(out, dtype=np.int64)
