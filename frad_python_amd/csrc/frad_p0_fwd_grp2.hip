// frad_p0_fwd_grp2.hip -- profile 0 encode of frames whose float64 channels need exactly two passes through a CU's LDS
// (C = 2 * CG; e.g. 8-channel integer PCM at N = 4096: 8 x 32 KiB), with whole-row global I/O: the encode-side twin of
// frad_p0_inv_grp2.hip.
//
// The generic channel-group kernel (k_p0_fwd_grp) stages and packs each pass's share of every row on its own: half rows,
// element-sized accesses, twice the load latency exposures.  Here a lane loads WHOLE PCM rows once (the second group's
// raw half waits in registers while the first group is transformed), keeps the first group's storage codes in registers
// while the second group is transformed, and then stores whole payload rows with 16-byte stores.
// Reference: fourier/profile0.py:14-44 (analogue), backend/pcmformat.py:34-47 (to_f64).
#include "frad_launch.hpp"

namespace frad {

// storage code of value v (channel j of the row) -> its bytes inside the row's word array
template <int BITS, int NW>
__device__ __forceinline__ void put_code(uint32_t (&w)[NW], int j, u64 code, bool le) {
    if constexpr (BITS == 16) { uint32_t c = (uint32_t)code & 0xffffu; if (!le) c = bswap16(c); w[j >> 1] |= c << (16 * (j & 1)); }
    else if constexpr (BITS == 32) { w[j] = le ? (uint32_t)code : bswap32((uint32_t)code); }
    else { const u64 c = le ? code : bswap64(code); w[2 * j] = (uint32_t)c; w[2 * j + 1] = (uint32_t)(c >> 32); }
}

template <int LOG2M, int CG, int LG, int BITS, int NH = 1>
__global__ void __launch_bounds__(CG * Plan<LOG2M>::TEAM, NH)   // NH = 2: 256 threads, two blocks per CU -> 2 waves per SIMD
k_p0_fwd_grp2(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH, SLOTS = padded_slots(M);
    constexpr int T = CG * TEAM, RPT = N / T, C = 2 * CG * NH, NBV = BITS / 8;
    constexpr int IW = (CG << LG) / 4;                        // words of half a PCM row
    constexpr int OW = CG * NBV / 4, ROW_OUT = C * NBV;       // words of a group's share of a payload row; bytes of a whole row
    static_assert(N % T == 0 && ((CG << LG) % 4) == 0 && IW >= 1 && (CG * NBV) % 4 == 0 && (2 * OW) % 4 == 0, "whole 16-byte pieces per block and row");
    FRAD_DYN_SMEM(smem);
    // NH = 2: a frame's rows are shared by two blocks (2 x CG channels each); blocks b and b + 8 run on the same XCD
    // (round-robin dispatch) and read / write their halves of every row through one L2 (frad_p0_inv_grp2.hip)
    long long f = blockIdx.x;
    int part = 0;
    if constexpr (NH == 2) { const long long r = f >> 3; part = (int)(r & 1); f = (r >> 1) * 8 + (f & 7); if (f >= g.n_frames) return; }
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const bool le = g.le != 0;
    const unsigned char* src = pcm + ((f * g.frame_stride * C) << LG) + part * (2 * IW * 4);
    constexpr int ROW_IN = C << LG;
    uint32_t hold[RPT][IW];
    u64 mx = 0;
    dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
        constexpr int CODE = decltype(code_tag)::value;
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        auto elem = [&](const uint32_t (&w)[IW], int j) -> double { return cvt_pcm_c<double, CODE, RAW>(word_elem<LG>(w, j)); };
        // ---- group 0 from memory, group 1's half of every PCM row parked in registers
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int n = threadIdx.x + i * T;
            uint32_t w[IW], both[2 * IW];                     // this block's share of the row in one access
            load_words<2 * IW>(src + (long long)n * ROW_IN, both);
#pragma unroll
            for (int q = 0; q < IW; ++q) { w[q] = both[q]; hold[i][q] = both[IW + q]; }
            const int m = makhoul(n, N);
#pragma unroll
            for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, m) = elem(w, j);
        }
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);
        fft_team<double, LOG2M, false>(buf, tt, tw);
        dct_post<double, LOG2M>(buf, tt, post);
        __syncthreads();
        // ---- group 0's storage codes into registers (as the first half of every payload row)
        uint32_t half0[RPT][OW];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int k = threadIdx.x + i * T;
#pragma unroll
            for (int q = 0; q < OW; ++q) half0[i][q] = 0;
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const double v = xslot<double, SH>(smem, j, SLOTS, k);
                const u64 a = abs_bits(v); mx = a > mx ? a : mx;
                put_code<BITS>(half0[i], j, storage_code<double>(v, BITS), le);
            }
        }
        __syncthreads();
        // ---- group 1 from the registers
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int m = makhoul(threadIdx.x + i * T, N);
#pragma unroll
            for (int j = 0; j < CG; ++j) xslot<double, SH>(smem, j, SLOTS, m) = elem(hold[i], j);
        }
        __syncthreads();
        tt = t; FRAD_OPAQUE(tt);
        fft_team<double, LOG2M, false>(buf, tt, tw);
        dct_post<double, LOG2M>(buf, tt, post);
        __syncthreads();
        // ---- whole payload rows out
        unsigned char* dst = payload + f * g.payload_stride + part * (2 * OW * 4);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int k = threadIdx.x + i * T;
            uint32_t row[2 * OW];
#pragma unroll
            for (int q = 0; q < OW; ++q) { row[q] = half0[i][q]; row[OW + q] = 0; }
            uint32_t h1[OW];
#pragma unroll
            for (int q = 0; q < OW; ++q) h1[q] = 0;
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const double v = xslot<double, SH>(smem, j, SLOTS, k);
                const u64 a = abs_bits(v); mx = a > mx ? a : mx;
                put_code<BITS>(h1, j, storage_code<double>(v, BITS), le);
            }
#pragma unroll
            for (int q = 0; q < OW; ++q) row[OW + q] = h1[q];
            store_words<2 * OW>(dst + (long long)k * ROW_OUT, row);
        }
    });
    if (absmax != nullptr) {                                  // np.max(np.abs(freqs)) of the frame (no NaN can arise: integer / finite input
        mx = wave_max_u64(mx);                                //  is required by the launcher for this kernel)
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<u64*>(absmax) + f, mx);
    }
}

namespace {
template <int LOG2M, int CG, int LG, int NH = 1>
int go_bits(int bits, size_t lds, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
            const cx<double>* tw, const cx<double>* post, const Geom& g) {
    constexpr int T = CG * Plan<LOG2M>::TEAM;
#define GO(B) do { allow_lds(k_p0_fwd_grp2<LOG2M, CG, LG, B, NH>, lds); \
        hipLaunchKernelGGL((k_p0_fwd_grp2<LOG2M, CG, LG, B, NH>), grid, dim3(T), lds, s, pcm, pay, am, tw, post, g); } while (0)
    if (bits == 32) { GO(32); return 1; }
    if (bits == 64) { GO(64); return 1; }
    if constexpr ((2 * CG * 2) % 16 == 0) { if (bits == 16) { GO(16); return 1; } }   // (store_words moves 16-byte pieces)
#undef GO
    return 0;
}
template <int LOG2M, int CG, int NH = 1>
int go_lg(int lg, int bits, size_t lds, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
          const cx<double>* tw, const cx<double>* post, const Geom& g) {
    if (lg == 1) { if constexpr ((CG << 1) % 4 == 0) return go_bits<LOG2M, CG, 1, NH>(bits, lds, grid, s, pcm, pay, am, tw, post, g); else return 0; }
    if (lg == 2) return go_bits<LOG2M, CG, 2, NH>(bits, lds, grid, s, pcm, pay, am, tw, post, g);
    return 0;                                                 // 8-byte PCM: 64 registers of parked input per lane -- the generic kernel
}
}  // namespace

// 1 = launched; 0 = not this kernel's geometry.  Needs 16-byte aligned PCM rows and payload rows, integer PCM of 2 or 4
// bytes (float64 PCM could carry NaN into the maximum; float32 PCM is transformed in float32 elsewhere), whole frames.
int launch_p0_fwd_grp2(int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
                       const Tables& tb, const Geom& g, int ai, int ao) {
    if (tune("FRAD_TUNE_NO_GRP2")) return 0;
    if (g.C != 2 * c.cg || g.n_frames > 0x7fffffffLL || !ai || !ao || g.n_valid != g.N || (g.dtype >> 3) == 2) return 0;
    if ((((long long)g.C) << lg) % 4 != 0 || (g.C * (g.bits / 8)) % 16 != 0) return 0;
    const cx<double>* tw = static_cast<const cx<double>*>(tb.tw);
    const cx<double>* post = static_cast<const cx<double>*>(tb.post);
    dim3 grid((unsigned)g.n_frames);
    // two half-size blocks per frame, two resident per CU (see launch_p0_inv_grp2)
    if (!tune("FRAD_TUNE_GRP2_WHOLE")) {
        const long long nb = ((g.n_frames + 7) / 8) * 16;
        if (nb <= 0x7fffffffLL) {
            const dim3 grid2((unsigned)nb);
            if (c.log2m == 10 && c.cg == 8 && go_lg<10, 4, 2>(lg, g.bits, c.lds / 2, grid2, s, pcm, pay, am, tw, post, g)) return 1;
            if (c.log2m == 11 && c.cg == 4 && go_lg<11, 2, 2>(lg, g.bits, c.lds / 2, grid2, s, pcm, pay, am, tw, post, g)) return 1;
        }
    }
    if (c.log2m == 10 && c.cg == 8) return go_lg<10, 8>(lg, g.bits, c.lds, grid, s, pcm, pay, am, tw, post, g);
    if (c.log2m == 11 && c.cg == 4) return go_lg<11, 4>(lg, g.bits, c.lds, grid, s, pcm, pay, am, tw, post, g);
    if (c.log2m == 12 && c.cg == 2) return go_lg<12, 2>(lg, g.bits, c.lds, grid, s, pcm, pay, am, tw, post, g);
    return 0;
}

}  // namespace frad
