"""Validation tables of the transform core (mirror of src/libfrad/fourier/__init__.py:3-25).

Profile 2 ("in development" upstream, not in AVAILABLE there either) is not built."""
from . import profiles

AVAILABLE = [0, 1, 4]
SEGMAX = [0xFFFFFFFF, profiles.compact.MAX_SMPL, profiles.compact.MAX_SMPL, 0, 0xFFFFFFFF, 0, 0, 0]
BIT_DEPTHS = [(12, 16, 24, 32, 48, 64), (8, 12, 16, 24, 32, 48, 64), (), (), (12, 16, 24, 32, 48, 64), (), (), ()]
