"""Profile classes and the compact-profile tables.

Same names as the reference (`LOSSLESS`, `COMPACT`, `compact.SRATES`, `compact.SAMPLES`,
`compact.get_*`; src/libfrad/fourier/profiles.py) because the streaming classes and the ASFH header
index into them; the legal frame sizes are generated ({128,160,192,224} << n) and looked up by bisection."""
from bisect import bisect_left

LOSSLESS = [0, 4]          # DCT archiving, PCM archiving
COMPACT = [1, 2]           # psychoacoustic profiles (2 is not built upstream either)

_RATES_DESC = (96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000)
_RATES_ASC = tuple(sorted(_RATES_DESC))
_SIZES = tuple(base << shift for shift in range(8) for base in (128, 160, 192, 224))


def _first_at_least(table, value, what):
    i = bisect_left(table, value)
    if i == len(table):
        raise ValueError(f"no legal compact {what} >= {value}")
    return table[i]


class compact:
    SRATES = _RATES_DESC                       # header index order (descending)
    SAMPLES = list(_SIZES)                     # header index order (ascending)
    MAX_SMPL = _SIZES[-1]

    @staticmethod
    def get_valid_srate(srate: int) -> int:
        return _first_at_least(_RATES_ASC, srate, "sample rate")

    @staticmethod
    def get_srate_index(srate: int) -> int:
        return _RATES_DESC.index(compact.get_valid_srate(srate))

    @staticmethod
    def get_samples_min_ge(smpl: int) -> int:
        return _first_at_least(_SIZES, smpl, "frame size")

    @staticmethod
    def get_samples_index(smpl: int) -> int:
        return bisect_left(_SIZES, compact.get_samples_min_ge(smpl))
