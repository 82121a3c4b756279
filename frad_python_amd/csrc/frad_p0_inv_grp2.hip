// frad_p0_inv_grp2.hip -- profile 0 decode of frames whose float64 channels need exactly two passes through a
// CU's LDS (C = 2 * CG; e.g. cfg 4: 192 kHz 7.1, N = 4096, 8 x 32 KiB), with whole-row global I/O.
//
// The generic channel-group kernel (k_p0_inv_grp) reads and writes each pass's share of every row on its
// own: half rows, twice the load latency exposures, and only element-sized accesses in flight -- 0.6 TB/s
// on the unpack alone.  Here a lane loads WHOLE payload rows once (the second group's half waits in
// registers while the first group is transformed), keeps the first group's time samples in registers while
// the second group is transformed, and then stores whole interleaved output rows: every payload byte is
// read once and every output row is written with contiguous 16-byte stores.
#include "frad_launch.hpp"

#include <cstdlib>

namespace frad {

template <int BITS, int NW>
__device__ __forceinline__ double piece_value(const uint32_t (&w)[NW], int v, bool le) {
    u64 c;
    if constexpr (BITS == 16) { c = (w[v >> 1] >> (16 * (v & 1))) & 0xffffu; if (!le) c = bswap16((uint32_t)c); }
    else if constexpr (BITS == 32) { c = w[v]; if (!le) c = bswap32((uint32_t)c); }
    else { c = (u64)w[2 * v] | ((u64)w[2 * v + 1] << 32); if (!le) c = bswap64(c); }
    return code_to_f64(c, BITS);
}

template <int LOG2M, int CG, int BITS>
__global__ void __launch_bounds__(CG * Plan<LOG2M>::TEAM)
k_p0_inv_grp2(const unsigned char* __restrict__ payload, double* __restrict__ out,
              const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH, SLOTS = padded_slots(M);
    constexpr int T = CG * TEAM, RPT = N / T, NBV = BITS / 8, C = 2 * CG, PW = CG * NBV / 4;   // words per half row
    static_assert(N % T == 0 && (CG * NBV) % 4 == 0 && PW >= 1, "whole words per half row");
    FRAD_DYN_SMEM(smem);
    const long long f = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const bool le = g.le != 0;
    const unsigned char* src = payload + f * g.payload_stride;
    constexpr int ROWB = C * NBV;
    uint32_t hold[RPT][PW];
    // ---- group 0 from memory, group 1's half of every row parked in registers
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int k = threadIdx.x + i * T;
        uint32_t w[PW];
        load_words<PW>(src + (long long)k * ROWB, w);
        load_words<PW>(src + (long long)k * ROWB + PW * 4, hold[i]);
#pragma unroll
        for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, k) = piece_value<BITS>(w, j, le);
    }
    __syncthreads();
    int tt = t; FRAD_OPAQUE(tt);
    dct_pre_inverse<double, LOG2M>(buf, tt, post);
    fft_team<double, LOG2M, true>(buf, tt, tw);
    __syncthreads();
    double res[RPT][CG];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int m = makhoul(threadIdx.x + i * T, N);
#pragma unroll
        for (int j = 0; j < CG; ++j) res[i][j] = xslot<double, SH>(smem, j, SLOTS, m);
    }
    __syncthreads();
    // ---- group 1 from the registers
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int k = threadIdx.x + i * T;
#pragma unroll
        for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, k) = piece_value<BITS>(hold[i], j, le);
    }
    __syncthreads();
    tt = t; FRAD_OPAQUE(tt);
    dct_pre_inverse<double, LOG2M>(buf, tt, post);
    fft_team<double, LOG2M, true>(buf, tt, tw);
    __syncthreads();
    double* dst = out + f * (long long)N * C;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int n = threadIdx.x + i * T, m = makhoul(n, N);
        double* row = dst + (long long)n * C;
        if constexpr (CG >= 2) {
#pragma unroll
            for (int j = 0; j < CG; j += 2) { v2d v = {res[i][j], res[i][j + 1]}; *FRAD_GPTR(v2d, row + j) = v; }
#pragma unroll
            for (int j = 0; j < CG; j += 2) {
                v2d v = {xslot<double, SH>(smem, j, SLOTS, m), xslot<double, SH>(smem, j + 1, SLOTS, m)};
                *FRAD_GPTR(v2d, row + CG + j) = v;
            }
        } else {
            v2d v = {res[i][0], xslot<double, SH>(smem, 0, SLOTS, m)};
            *FRAD_GPTR(v2d, row) = v;
        }
    }
}

template <int LOG2M, int CG>
int go_grp2(int bits, size_t lds, dim3 grid, hipStream_t s, const unsigned char* pay, double* out, const cx<double>* tw,
            const cx<double>* post, const Geom& g) {
    constexpr int T = CG * Plan<LOG2M>::TEAM;
#define GO(B) do { allow_lds(k_p0_inv_grp2<LOG2M, CG, B>, lds); \
        hipLaunchKernelGGL((k_p0_inv_grp2<LOG2M, CG, B>), grid, dim3(T), lds, s, pay, out, tw, post, g); } while (0)
    if (bits == 32) { GO(32); return 1; }
    if (bits == 64) { GO(64); return 1; }
    if constexpr (CG >= 2) { if (bits == 16) { GO(16); return 1; } }
#undef GO
    return 0;
}

// 1 = launched; 0 = not this kernel's geometry
int launch_p0_inv_grp2(const FastCfg& c, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb, const Geom& g) {
    if (tune("FRAD_TUNE_NO_GRP2")) return 0;                                       // A/B knob, not part of the ABI
    if (g.C != 2 * c.cg || g.n_frames > 0x7fffffffLL) return 0;
    if ((reinterpret_cast<uintptr_t>(pay) & 15) || (g.payload_stride & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return 0;
    const cx<double>* tw = static_cast<const cx<double>*>(tb.tw);
    const cx<double>* post = static_cast<const cx<double>*>(tb.post);
    dim3 grid((unsigned)g.n_frames);
    if (c.log2m == 10 && c.cg == 8) return go_grp2<10, 8>(g.bits, c.lds, grid, s, pay, out, tw, post, g);
    if (c.log2m == 11 && c.cg == 4) return go_grp2<11, 4>(g.bits, c.lds, grid, s, pay, out, tw, post, g);
    if (c.log2m == 12 && c.cg == 2) return go_grp2<12, 2>(g.bits, c.lds, grid, s, pay, out, tw, post, g);
    if (c.log2m == 13 && c.cg == 1) return go_grp2<13, 1>(g.bits, c.lds, grid, s, pay, out, tw, post, g);
    return 0;
}

}  // namespace frad
