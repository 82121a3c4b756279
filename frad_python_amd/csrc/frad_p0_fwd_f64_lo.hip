// k_p0_fwd, float64 compute (integer and f64 PCM), N = 128 .. 1024
#define FWD_T double
#define FWD_NAME launch_p0_fwd_f64_lo
#define FWD_LO 6
#define FWD_HI 9
#include "frad_p0_fwd.inc"
