// k_p0_fwd, float64 compute (integer and f64 PCM), N = 2048 .. 16384
#define FWD_T double
#define FWD_NAME launch_p0_fwd_f64_hi
#define FWD_LO 10
#define FWD_HI 13
#include "frad_p0_fwd.inc"
