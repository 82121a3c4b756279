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

static int grp2_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        return 256;
    }();
    return n;
}

template <int BITS, int NW>
__device__ __forceinline__ double piece_value(const uint32_t (&w)[NW], int v, bool le) {
    u64 c;
    if constexpr (BITS == 16) { c = (w[v >> 1] >> (16 * (v & 1))) & 0xffffu; if (!le) c = bswap16((uint32_t)c); }
    else if constexpr (BITS == 32) { c = w[v]; if (!le) c = bswap32((uint32_t)c); }
    else { c = (u64)w[2 * v] | ((u64)w[2 * v + 1] << 32); if (!le) c = bswap64(c); }
    return code_to_f64(c, BITS);
}

// CONV: the samples leave in the caller's PCM format (Geom::dtype) -- an instantiation of its own, so that the float64 kernel of
// the BASELINE path keeps its registers (the converting store as a run-time branch cost it 164 B of scratch per lane)
template <int LOG2M, int CG, int BITS, int NH = 1, bool CONV = false>
__global__ void __launch_bounds__(CG * Plan<LOG2M>::TEAM, NH)   // NH = 2: 256 threads, two blocks per CU -> 2 waves per SIMD
k_p0_inv_grp2(const unsigned char* __restrict__ payload, double* __restrict__ out,
              const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH, SLOTS = padded_slots(M);
    constexpr int T = CG * TEAM, RPT = N / T, NBV = BITS / 8, C = 2 * CG * NH, PW = CG * NBV / 4;   // words per group and row
    static_assert(N % T == 0 && (CG * NBV) % 4 == 0 && PW >= 1, "whole words per half row");
    FRAD_DYN_SMEM(smem);
    // NH = 2: a frame's rows are shared by two blocks (2 x CG channels each).  Blocks b and b + 8 land on the same XCD
    // (round-robin dispatch), so the partners read their halves of every payload sector through one L2.
    long long f = blockIdx.x;
    int part = 0;
    if constexpr (NH == 2) { const long long r = f >> 3; part = (int)(r & 1); f = (r >> 1) * 8 + (f & 7); if (f >= g.n_frames) return; }
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const bool le = g.le != 0;
    const unsigned char* src = payload + f * g.payload_stride + part * (2 * PW * 4);
    constexpr int ROWB = C * NBV;
    uint32_t hold[RPT][PW];
    // ---- group 0 from memory, group 1's half of every row parked in registers
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int k = threadIdx.x + i * T;
        uint32_t w[PW];
#if defined(FRAD_X_GRP2) && (FRAD_X_GRP2 & 4)
        for (int q = 0; q < PW; ++q) { w[q] = 0x3c000000u + k * 64 + q; hold[i][q] = 0x3c100000u + k * 64 + q; }
#else
        uint32_t both[2 * PW];                                // this block's share of the row in one access
        load_words<2 * PW>(src + (long long)k * ROWB, both);
#pragma unroll
        for (int q = 0; q < PW; ++q) { w[q] = both[q]; hold[i][q] = both[PW + q]; }
#endif
#pragma unroll
        for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, k) = piece_value<BITS>(w, j, le);
    }
    __syncthreads();
    int tt = t; FRAD_OPAQUE(tt);
#if !(defined(FRAD_X_GRP2) && (FRAD_X_GRP2 & 1))
    dct_pre_inverse<double, LOG2M>(buf, tt, post);
    fft_team<double, LOG2M, true>(buf, tt, tw);
#endif
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
#if !(defined(FRAD_X_GRP2) && (FRAD_X_GRP2 & 1))
    dct_pre_inverse<double, LOG2M>(buf, tt, post);
    fft_team<double, LOG2M, true>(buf, tt, tw);
#endif
    __syncthreads();
    if constexpr (CONV) {                                     // the caller's PCM format instead of float64 (frad_p0_digital_pcm;
        dispatch_out_format(g.dtype, [&](auto kind_tag, auto lg_tag) {            //  backend/pcmformat.py:49-62 applied in the store)
            constexpr int KIND = decltype(kind_tag)::value, LGS = decltype(lg_tag)::value;
            const bool be = (g.dtype & 1) != 0, raw = g.raw_be != 0 && be;
            unsigned char* dstb = reinterpret_cast<unsigned char*>(out) + ((f * (long long)N * C + part * (2 * CG)) << LGS);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int n = threadIdx.x + i * T, m = makhoul(n, N);
                unsigned char* row = dstb + (((long long)n * C) << LGS);
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    store_pcm_elem<LGS>(row + (j << LGS), from_f64_bits<KIND, LGS>(res[i][j], raw), be);
                    store_pcm_elem<LGS>(row + ((CG + j) << LGS), from_f64_bits<KIND, LGS>(xslot<double, SH>(smem, j, SLOTS, m), raw), be);
                }
            }
        });
        return;
    }
    double* dst = out + f * (long long)N * C + part * (2 * CG);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int n = threadIdx.x + i * T, m = makhoul(n, N);
        double* row = dst + (long long)n * C;
#if defined(FRAD_X_GRP2) && (FRAD_X_GRP2 & 2)
        if (res[i][0] != 1.2345e-300 || xslot<double, SH>(smem, 1, SLOTS, m) != 1.2345e-300) continue;
#endif
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

// The same two passes as a persistent, software-pipelined loop over frames (grid = CUs; one 128 KiB block per CU cannot
// overlap its own phases any other way): each group's samples leave as half rows (CG doubles = whole 32-byte sectors for
// CG >= 4) straight after its transform, so the stores drain under the next transform, and the next frame's payload rows
// are fetched into the registers the current frame's rows have just left, under the second transform.
template <int LOG2M, int CG, int BITS>
__global__ void __launch_bounds__(CG * Plan<LOG2M>::TEAM)
k_p0_inv_grp2p(const unsigned char* __restrict__ payload, double* __restrict__ out,
               const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH, SLOTS = padded_slots(M);
    constexpr int T = CG * TEAM, RPT = N / T, NBV = BITS / 8, C = 2 * CG, PW = CG * NBV / 4;   // words per half row
    static_assert(N % T == 0 && (CG * NBV) % 4 == 0 && PW >= 1 && CG >= 2, "whole words per half row");
    FRAD_DYN_SMEM(smem);
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const bool le = g.le != 0;
    constexpr int ROWB = C * NBV;
    uint32_t hold[RPT][2 * PW];
    auto fetch = [&](long long f) {
        const unsigned char* src = payload + f * g.payload_stride;
#pragma unroll
        for (int i = 0; i < RPT; ++i) load_words<2 * PW>(src + (long long)(threadIdx.x + i * T) * ROWB, hold[i]);
    };
    auto stage = [&](int half) {                              // one group's storage codes -> LDS as float64
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int k = threadIdx.x + i * T;
            uint32_t w[PW];
#pragma unroll
            for (int q = 0; q < PW; ++q) w[q] = hold[i][half * PW + q];
#pragma unroll
            for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, k) = piece_value<BITS>(w, j, le);
        }
    };
    auto drain = [&](double* dst, int half) {                 // one group's samples -> its half of every output row
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int n = threadIdx.x + i * T, m = makhoul(n, N);
            double* row = dst + (long long)n * C + half * CG;
#pragma unroll
            for (int j = 0; j < CG; j += 2) {
                v2d v = {xslot<double, SH>(smem, j, SLOTS, m), xslot<double, SH>(smem, j + 1, SLOTS, m)};
                *FRAD_GPTR(v2d, row + j) = v;
            }
        }
    };
    long long f = blockIdx.x;
    if (f < g.n_frames) fetch(f);
    for (; f < g.n_frames; f += gridDim.x) {
        double* dst = out + f * (long long)N * C;
        stage(0);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);
        dct_pre_inverse<double, LOG2M>(buf, tt, post);
        fft_team<double, LOG2M, true>(buf, tt, tw);
        __syncthreads();
        drain(dst, 0);
        __syncthreads();
        stage(1);
        if (f + gridDim.x < g.n_frames) fetch(f + gridDim.x);
        __syncthreads();
        tt = t; FRAD_OPAQUE(tt);
        dct_pre_inverse<double, LOG2M>(buf, tt, post);
        fft_team<double, LOG2M, true>(buf, tt, tw);
        __syncthreads();
        drain(dst, 1);
        __syncthreads();
    }
}

template <int LOG2M, int CG, int NH = 1>
int go_grp2(int bits, size_t lds, dim3 grid, int pipe_grid, hipStream_t s, const unsigned char* pay, double* out,
            const cx<double>* tw, const cx<double>* post, const Geom& g) {
    constexpr int T = CG * Plan<LOG2M>::TEAM;
#define GO(B) do { if (g.dtype != 22) { allow_lds(k_p0_inv_grp2<LOG2M, CG, B, NH, true>, lds); \
        hipLaunchKernelGGL((k_p0_inv_grp2<LOG2M, CG, B, NH, true>), grid, dim3(T), lds, s, pay, out, tw, post, g); } else { \
        allow_lds(k_p0_inv_grp2<LOG2M, CG, B, NH>, lds); \
        hipLaunchKernelGGL((k_p0_inv_grp2<LOG2M, CG, B, NH>), grid, dim3(T), lds, s, pay, out, tw, post, g); } } while (0)
    if constexpr (CG >= 2 && NH == 1) {
        if (pipe_grid > 0) {
#define GOP(B) do { allow_lds(k_p0_inv_grp2p<LOG2M, CG, B>, lds); \
        hipLaunchKernelGGL((k_p0_inv_grp2p<LOG2M, CG, B>), dim3(pipe_grid), dim3(T), lds, s, pay, out, tw, post, g); } while (0)
            if (bits == 32) { GOP(32); return 1; }
            if (bits == 64) { GOP(64); return 1; }
            if (bits == 16) { GOP(16); return 1; }
#undef GOP
            return 0;
        }
    }
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
    const bool conv = g.dtype != 22;                          // not FRAD_PCM_F64LE: the store converts (element stores: no alignment asked of `out`)
    if ((reinterpret_cast<uintptr_t>(pay) & 15) || (g.payload_stride & 15) || (!conv && (reinterpret_cast<uintptr_t>(out) & 15))) return 0;
    const cx<double>* tw = static_cast<const cx<double>*>(tb.tw);
    const cx<double>* post = static_cast<const cx<double>*>(tb.post);
    dim3 grid((unsigned)g.n_frames);
    // persistent pipelined variant: one block per CU, each walking frames blockIdx.x, + grid, ...
    int pipe = 0;
    if (!conv && tune("FRAD_TUNE_GRP2_PIPE")) { const int cus = grp2_cu_count(); pipe = (int)(g.n_frames < cus ? g.n_frames : cus); }
    // Two half-size blocks per frame (2 x CG/2 channels each), two blocks resident per CU: one block's loads and stores
    // run under the other's transforms (a single block serialises them: cfg 4, N = 4096 x 8 channels, 0.53 -> 0.45 ms)
    if (!tune("FRAD_TUNE_GRP2_WHOLE")) {
        const long long nb = ((g.n_frames + 7) / 8) * 16;
        if (nb <= 0x7fffffffLL) {
            const dim3 grid2((unsigned)nb);
            if (c.log2m == 10 && c.cg == 8 && go_grp2<10, 4, 2>(g.bits, c.lds / 2, grid2, 0, s, pay, out, tw, post, g)) return 1;
            if (c.log2m == 11 && c.cg == 4 && go_grp2<11, 2, 2>(g.bits, c.lds / 2, grid2, 0, s, pay, out, tw, post, g)) return 1;
            // (N = 8192 x 4 channels as 2 x (1 + 1) is slower than one 2 + 2 block: 0.58 against 0.50 ms -- 8-byte loads)
        }
    }
    if (c.log2m == 10 && c.cg == 8) return go_grp2<10, 8>(g.bits, c.lds, grid, pipe, s, pay, out, tw, post, g);
    if (c.log2m == 11 && c.cg == 4) return go_grp2<11, 4>(g.bits, c.lds, grid, pipe, s, pay, out, tw, post, g);
    if (c.log2m == 12 && c.cg == 2) return go_grp2<12, 2>(g.bits, c.lds, grid, pipe, s, pay, out, tw, post, g);
    if (c.log2m == 13 && c.cg == 1) return go_grp2<13, 1>(g.bits, c.lds, grid, 0, s, pay, out, tw, post, g);
    return 0;
}

}  // namespace frad
