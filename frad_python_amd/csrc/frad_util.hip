// frad_util.hip -- measurement aid: a plain device-to-device copy with the access shape of the transform kernels
// (16 bytes per lane, fully coalesced, grid-stride), so that "achievable HBM bandwidth" in bench.py / DESIGN.md is
// measured with the same instruction mix as the product instead of borrowed from a library memcpy.
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

namespace frad {

__global__ void __launch_bounds__(256) k_bench_copy(const v4u* __restrict__ src, v4u* __restrict__ dst, long long n16) {
    // one 16 KiB tile per block and step: four 16-byte loads in flight per lane, every wave instruction 1 KiB contiguous
    const long long tiles = (n16 + 1023) / 1024;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long i = t * 1024 + threadIdx.x;
        if (i + 768 < n16) {
            const v4u a = FRAD_NT_LOAD(FRAD_GCPTR(v4u, src) + i), b = FRAD_NT_LOAD(FRAD_GCPTR(v4u, src) + i + 256),
                      c = FRAD_NT_LOAD(FRAD_GCPTR(v4u, src) + i + 512), d = FRAD_NT_LOAD(FRAD_GCPTR(v4u, src) + i + 768);
            FRAD_NT_STORE(a, FRAD_GPTR(v4u, dst) + i); FRAD_NT_STORE(b, FRAD_GPTR(v4u, dst) + i + 256);
            FRAD_NT_STORE(c, FRAD_GPTR(v4u, dst) + i + 512); FRAD_NT_STORE(d, FRAD_GPTR(v4u, dst) + i + 768);
        } else {
            for (long long j = i; j < n16 && j < (t + 1) * 1024; j += 256) FRAD_GPTR(v4u, dst)[j] = FRAD_GCPTR(v4u, src)[j];
        }
    }
}

}  // namespace frad

extern "C" int frad_bench_copy(const void* src, void* dst, int64_t nbytes, void* stream) {
    if (nbytes < 0 || (nbytes & 15) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return FRAD_E_INVALID;
    if (nbytes == 0) return FRAD_OK;
    if (!src || !dst) return FRAD_E_INVALID;
    const long long n16 = nbytes / 16;
    long long blocks = (n16 + 1023) / 1024;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(frad::k_bench_copy, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const frad::v4u*>(src), static_cast<frad::v4u*>(dst), n16);
    return hipGetLastError() == hipSuccess ? FRAD_OK : FRAD_E_HIP;
}
