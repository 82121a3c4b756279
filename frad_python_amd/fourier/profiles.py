"""Profile classes and the compact-profile tables (mirror of src/libfrad/fourier/profiles.py)."""
LOSSLESS = [0, 4]
COMPACT = [1, 2]


class compact:
    SRATES = (96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000)
    SAMPLES = [m << s for s in range(8) for m in (128, 160, 192, 224)]
    MAX_SMPL = max(SAMPLES)

    @staticmethod
    def get_valid_srate(srate: int) -> int:
        return min(x for x in compact.SRATES if x >= srate)

    @staticmethod
    def get_srate_index(srate: int) -> int:
        return compact.SRATES.index(compact.get_valid_srate(srate))

    @staticmethod
    def get_samples_min_ge(smpl: int) -> int:
        return min(x for x in compact.SAMPLES if x >= smpl)

    @staticmethod
    def get_samples_index(smpl: int) -> int:
        return compact.SAMPLES.index(compact.get_samples_min_ge(smpl))
