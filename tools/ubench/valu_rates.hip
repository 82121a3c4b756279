// Micro-benchmark (diagnostic, not part of the library): issue cost of the VALU / LDS instructions the FFT kernels use,
// in shader cycles per wave-instruction, at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITER 200
template <int OP> __global__ void __launch_bounds__(512) k(double* out, unsigned long long* cyc, int n) {
    __shared__ double lds[4096];
    double a[8], b = threadIdx.x * 1e-3 + 1.0, c = 0.5;
    int ia[8]; float fa[8];
    for (int i = 0; i < 8; ++i) { a[i] = b + i; ia[i] = threadIdx.x + i; fa[i] = (float)i; }
    lds[threadIdx.x] = b; lds[threadIdx.x + 512] = b;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if constexpr (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                else if constexpr (OP == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));
                else if constexpr (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[i]) : "v"(a[i]));
                else if constexpr (OP == 5) asm volatile("v_max_f64 %0, %0, |%1|" : "+v"(a[i]) : "v"(c));
                else if constexpr (OP == 6) asm volatile("v_bfe_i32 %0, %0, %1, 16" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                else if constexpr (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                else if constexpr (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                else if constexpr (OP == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fa[(i + 1) & 7]));
                else if constexpr (OP == 10) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(fa[i]));
                else if constexpr (OP == 11) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));
                else if constexpr (OP == 12) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(ia[i]));
                else if constexpr (OP == 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                else if constexpr (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                else if constexpr (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; int is = 0; float fs = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; is += ia[i]; fs += fa[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + is + fs;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char* name) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, ITER);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, ITER);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256 * (threads / 64); ++i) m += h[i];
        m /= 256 * (threads / 64);
        printf("%-16s %d waves/SIMD: %.2f cycles per wave-instruction (per wave), %.2f per SIMD\n", name, threads / 256, m / (ITER * REP), m / (ITER * REP) / (threads / 256));
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<3>("v_cvt_f64_i32"); run<4>("v_cvt_f32_f64");
    run<5>("v_max_f64"); run<6>("v_bfe_i32"); run<7>("v_perm_b32"); run<8>("v_cndmask_b32"); run<9>("v_add_f32");
    run<10>("v_cvt_f64_f32"); run<11>("v_cvt_f64_u32"); run<12>("v_ldexp_f64"); run<13>("v_xor_b32"); run<14>("v_mov_b32"); run<15>("v_lshl_add_u32");
    return 0;
}
