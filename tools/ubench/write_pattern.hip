// Micro-benchmark (diagnostic): HBM write throughput of two address schedules with the same bytes, same stores:
//   linear : at any instant the chip's waves write one contiguous window (a copy kernel's schedule)
//   framed : wave w walks ITS OWN 32 KiB frame 1 KiB at a time, all waves at the same offset inside their frames at
//            the same time (the schedule of a frame-per-wave codec kernel)
//   framed+rot : the same, each wave starting at a different 1 KiB piece of its frame
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((vector_size(16)));
template <int MODE> __global__ void __launch_bounds__(448) k(v4u* dst, long long frames, int frame_kib) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 7 + wv, nw = (long long)gridDim.x * 7;
    v4u val = {1u, 2u, 3u, (unsigned)w};
    if (MODE == 0) {
        const long long pieces = frames * frame_kib;
        for (long long p = w; p < pieces; p += nw) dst[p * 64 + lane] = val;
    } else {
        for (long long f = w; f < frames; f += nw)
            for (int c = 0; c < frame_kib; ++c) {
                const int cc = MODE == 2 ? (c + (int)(f * 5)) % frame_kib : c;
                dst[(f * frame_kib + cc) * 64 + lane] = val;
            }
    }
}
int main() {
    const long long frames = 14062; const int kib = 32;
    v4u* d; hipMalloc(&d, frames * kib * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(448), 0, 0, d, frames, kib);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(448), 0, 0, d, frames, kib);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(448), 0, 0, d, frames, kib);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("mode %d (%s): %.1f us per pass, %.2f TB/s of writes\n", mode, mode == 0 ? "linear" : mode == 1 ? "framed" : "framed+rot",
                                 ms / 20 * 1e3, frames * kib * 1024.0 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
