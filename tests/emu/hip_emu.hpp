// hip_emu.hpp -- TEST INFRASTRUCTURE: a thread-per-lane interpreter for the FrAD HIP kernels.
//
// Lets the build container (no GPU) execute frad_kernels.hpp / frad_hip.hip unchanged, compiled by
// g++ with -DFRAD_HOST_EMULATION, so that indexing, packing and barrier structure can be checked
// against the oracle and under AddressSanitizer.  One std::thread per HIP thread, one block at a
// time; __syncthreads is a std::barrier over the block, wave barriers and shuffles a std::barrier
// over the 64 lanes of the wave (lanes are NOT in lockstep here, so every wave-level exchange in
// the kernels must be fenced -- which the emulator therefore also checks).  Never linked into
// libfrad_hip.so.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 { unsigned x = 1, y = 1, z = 1; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) double2 { double x, y; };

namespace emu {
struct Block {
    std::barrier<> bar;                                       // block-wide (__syncthreads)
    std::vector<std::unique_ptr<std::barrier<>>> wave;        // one per 64-lane wave (wave-level ops)
    std::vector<unsigned long long> scratch;
    std::vector<unsigned char> smem;
    explicit Block(unsigned n, size_t lds) : bar(n), scratch(n), smem(lds + 64) {
        for (unsigned w = 0; w * 64 < n; ++w) wave.emplace_back(new std::barrier<>(std::min(64u, n - w * 64)));
    }
};
inline thread_local Block* blk = nullptr;
inline thread_local dim3 t_idx, b_idx, b_dim, g_dim;
inline unsigned char* smem_base() {
    auto p = reinterpret_cast<uintptr_t>(blk->smem.data());
    return reinterpret_cast<unsigned char*>((p + 15) & ~uintptr_t(15));
}
template <typename F> void launch(dim3 grid, dim3 block, size_t lds, F&& body) {
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        Block b(block.x, lds);
        std::vector<std::thread> th;
        th.reserve(block.x);
        for (unsigned tx = 0; tx < block.x; ++tx)
            th.emplace_back([&, tx] {
                blk = &b; t_idx = dim3(tx); b_idx = dim3(bx, by); b_dim = block; g_dim = grid;
                body();
            });
        for (auto& t : th) t.join();
    }
}
}  // namespace emu

#define threadIdx emu::t_idx
#define blockIdx emu::b_idx
#define blockDim emu::b_dim
#define gridDim emu::g_dim
#define FRAD_DYN_SMEM(name) unsigned char* name = emu::smem_base()
#define FRAD_OPAQUE(x) asm volatile("" : "+r"(x))
#define FRAD_PIN(x) ((void)(x))
#define FRAD_GPTR(T, p) ((T*)(p))
#define FRAD_GCPTR(T, p) ((const T*)(p))
#define FRAD_LDS_BARRIER() __syncthreads()

inline void __syncthreads() { emu::blk->bar.arrive_and_wait(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { emu::blk->wave[emu::t_idx.x / 64]->arrive_and_wait(); }

inline unsigned long long __shfl_xor(unsigned long long v, int mask, int /*width*/) {
    auto* b = emu::blk;
    auto& wb = *b->wave[emu::t_idx.x / 64];
    b->scratch[emu::t_idx.x] = v;
    wb.arrive_and_wait();
    const unsigned src = emu::t_idx.x ^ (unsigned)mask;                     // stays inside the wave (mask < 64)
    const unsigned long long r = src < emu::b_dim.x ? b->scratch[src] : v;
    wb.arrive_and_wait();
    return r;
}
inline unsigned long long __shfl(unsigned long long v, int src_lane, int /*width*/) {
    auto* b = emu::blk;
    auto& wb = *b->wave[emu::t_idx.x / 64];
    b->scratch[emu::t_idx.x] = v;
    wb.arrive_and_wait();
    const unsigned src = (emu::t_idx.x & ~63u) + (unsigned)src_lane;
    const unsigned long long r = src < emu::b_dim.x ? b->scratch[src] : v;
    wb.arrive_and_wait();
    return r;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; std::memcpy(&r, &v, 8); return r; }
inline uint32_t __float_as_uint(float f) { uint32_t r; std::memcpy(&r, &f, 4); return r; }
inline float __uint_as_float(uint32_t v) { float r; std::memcpy(&r, &v, 4); return r; }
using std::fma;

// float16 <-> float32, IEEE round-to-nearest-even
struct __half { unsigned short v; };
inline __half __ushort_as_half(unsigned short v) { return __half{v}; }
inline unsigned short __half_as_ushort(__half h) { return h.v; }
inline float __half2float(__half h) {
    const uint32_t s = (h.v & 0x8000u) << 16, e = (h.v >> 10) & 0x1f, m = h.v & 0x3ffu;
    if (e == 0) { const float v = std::ldexp((float)m, -24); return s ? -v : v; }
    if (e == 31) return __uint_as_float(s | 0x7f800000u | (m << 13));
    return __uint_as_float(s | ((e + 112) << 23) | (m << 13));
}
inline __half __float2half_rn(float f) {
    const uint32_t u = __float_as_uint(f), s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return __half{(unsigned short)(s | 0x7e00u | ((a >> 13) & 0x3ffu))};
    if (a >= 0x47800000u) return __half{(unsigned short)(s | 0x7c00u)};          // >= 65536 -> inf
    if (a < 0x33000000u) return __half{(unsigned short)s};                        // < 2^-25 -> 0
    const int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    uint32_t h = e < -14 ? q : (((uint32_t)(e + 15) << 10) + (q - 0x400u));       // carry propagates into the exponent
    return __half{(unsigned short)(s | h)};
}
inline float __double2float_rz(double d) {
    float r = (float)d;
    if (std::fabs((double)r) > std::fabs(d)) r = std::nextafterf(r, 0.0f);
    return r;
}

// ---- just enough of the HIP runtime for frad_hip.hip's host code -----------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void* hipStream_t;
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 3; return hipSuccess; }
template <typename P> inline hipError_t hipMalloc(P** p, size_t n) { *p = static_cast<P*>(std::aligned_alloc(64, (n + 63) / 64 * 64)); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
template <typename P> inline hipError_t hipMallocAsync(P** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, int, hipStream_t) {
    for (size_t r = 0; r < height; ++r) std::memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
struct hipDeviceProp_t { size_t maxSharedMemoryPerMultiProcessor = 163840, sharedMemPerBlockOptin = 163840; int multiProcessorCount = 1; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
struct hipFuncAttributes { int numRegs = 0; };
inline hipError_t hipFuncGetAttributes(hipFuncAttributes*, const void*) { return hipSuccess; }
template <typename K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
template <typename K, typename... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args) {
    emu::launch(grid, block, lds, [&] { kernel(args...); });
}
