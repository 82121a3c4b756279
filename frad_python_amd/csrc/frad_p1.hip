// profile 1 (psychoacoustic quantiser) entry points -- kernels K7/K8.
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

extern "C" {
int frad_p1_analogue(const void*, int32_t, int64_t, int32_t, int32_t, int64_t, int32_t, int32_t, int32_t, double,
                     uint32_t, int32_t*, int32_t*, void*) { return FRAD_E_UNSUPPORTED; }
int frad_p1_digital(const int32_t*, const int32_t*, int64_t, int32_t, int32_t, int32_t, int32_t, double*, void*) { return FRAD_E_UNSUPPORTED; }
int frad_p1_overlap_add(const double*, int64_t, int32_t, int32_t, int32_t, const double*, double*, double*, void*) { return FRAD_E_UNSUPPORTED; }
}
