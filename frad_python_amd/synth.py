"""Deterministic synthetic PCM used by the golden-vector generator, the tests and bench.py.

There is no dataset on the box; every workload of BASELINE.json is synthesised here from a
seed.  Pure NumPy, no device code.
"""
from __future__ import annotations

import numpy as np


def harmonic_mix(n: int, channels: int, srate: int, seed: int = 1234, noise_db: float = -60.0,
                 peak: float = 0.8) -> np.ndarray:
    """"Signal A": per channel c, 8 partials of 110*(c+1) Hz with 1/h roll-off and random
    phase, 0.5 Hz amplitude modulation, plus white noise at ``noise_db`` dBFS; float64 [n, C]
    normalised to ``peak``."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / srate
    out = np.empty((n, channels))
    for c in range(channels):
        f0 = 110.0 * (c + 1)
        ph = rng.uniform(0, 2 * np.pi, 8)
        x = sum(np.sin(2 * np.pi * f0 * (h + 1) * t + ph[h]) / (h + 1) for h in range(8))
        x *= 0.75 + 0.25 * np.sin(2 * np.pi * 0.5 * t + c)
        out[:, c] = x
    out *= peak / np.max(np.abs(out))
    out += rng.standard_normal((n, channels)) * 10.0 ** (noise_db / 20.0)
    return np.clip(out, -1.0, 1.0)


def uniform_full_scale(n: int, channels: int, seed: int = 7, amp: float = 1.0) -> np.ndarray:
    """"Signal B": i.i.d. uniform(-amp, amp) float64 [n, C] (worst case for absmax)."""
    return np.random.default_rng(seed).uniform(-amp, amp, (n, channels))


def sine(n: int, channels: int, srate: int, freq: float = 440.0, amp: float = 0.5) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / srate
    return np.repeat((amp * np.sin(2 * np.pi * freq * t))[:, None], channels, axis=1)


def to_pcm(x: np.ndarray, fmt: str) -> np.ndarray:
    """float64 in [-1, 1] -> array of the ffmpeg-style PCM format ``fmt`` (round to nearest,
    saturating) -- a synthesiser convenience, not the codec's ``from_f64``."""
    from .backend.pcmformat import ff_format_to_numpy_type
    dt = ff_format_to_numpy_type(fmt)
    if dt.kind == "f":
        return x.astype(dt)
    w = 8 * dt.itemsize
    half = float(1 << (w - 1))
    if dt.kind == "u":
        v = np.clip(np.rint((x + 1.0) * half), 0, 2.0 ** w - 1)
    else:
        v = np.clip(np.rint(x * half), -half, half - 1)
    return v.astype(dt)
