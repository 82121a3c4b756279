// k_p0_fwd, float32 compute (f32 / f16 PCM: the reference does not widen floats), all sizes
#define FWD_T float
#define FWD_NAME launch_p0_fwd_f32
#define FWD_LO 6
#define FWD_HI 13
#include "frad_p0_fwd.inc"
