"""frad_python_amd -- the MI355X-native FrAD transform core.

Drop-in for ONE hot path of H4n-uL/FrAD_Python (``src/libfrad``): the per-frame Fourier analysis /
synthesis, quantisation and bit-depth pack / unpack loop.  Same streaming API as the reference
(``Encoder.process / flush``, ``Decoder.process / flush``), same byte streams; the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C-ABI of ``include/frad_hip.h``.  Container metadata,
Reed-Solomon repair, the CLI and playback are out of scope.  There is no CPU fallback."""
from .fourier import AVAILABLE, BIT_DEPTHS, SEGMAX, profiles  # noqa: F401
from .backend.pcmformat import ff_format_to_numpy_type  # noqa: F401
from .tools.asfh import ASFH  # noqa: F401
from .encoder import Encoder, EncodeResult  # noqa: F401
from .decoder import Decoder, DecodeResult  # noqa: F401

__all__ = ["Encoder", "Decoder", "EncodeResult", "DecodeResult", "ASFH", "AVAILABLE", "BIT_DEPTHS", "SEGMAX", "profiles"]
