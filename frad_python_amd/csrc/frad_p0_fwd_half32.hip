// frad_p0_fwd_half32.hip -- profile 0 encode of float32 PCM frames with many channels (BASELINE config 4: 192 kHz 7.1,
// N = 4096, 8 channels, float32 compute as the reference does for float PCM: profile0.py:14-44, pcmformat.py:35).
//
// The persistent kernel these frames used (one 512-thread block per CU, one wave per channel, phases separated by block
// barriers) spends most of its time with either the memory system or the LDS idle: 0.34 ms per 2 812 frames = 0.27 of HBM.
// Here a frame's rows are shared by TWO blocks of C/2 channels each (the trick of frad_p0_inv_grp2.hip): a block loads its 16
// bytes of every 32-byte PCM row, transforms its four channels in ONE pass (float32: 4 x 16 KiB of LDS), and stores its 16
// bytes of every payload row.  Two such blocks are resident per CU (2 x 64 KiB, 16 waves), so one block's loads and stores
// run under the other's transform; partner blocks b and b + 8 land on the same XCD (round-robin dispatch) and meet in one L2.
// Tables come from L2 (the one-shot kernels' tables), not from LDS.
#include "frad_launch.hpp"

namespace frad {

template <int LOG2M, int CG, int BITS>
__global__ void __launch_bounds__(CG * Plan<LOG2M>::TEAM, 2)
k_p0_fwd_half32(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
                const cx<float>* __restrict__ tw, const cx<float>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH, SLOTS = padded_slots(M);
    constexpr int T = CG * TEAM, RPT = N / T, C = 2 * CG, NBV = BITS / 8;
    constexpr int IW = CG;                                    // words of this block's share of a PCM row (float32)
    constexpr int OW = CG * NBV / 4;                          // words of its share of a payload row
    constexpr int ROW_IN = C * 4, ROW_OUT = C * NBV;
    static_assert(N % T == 0 && IW % 4 == 0 && OW % 4 == 0, "whole 16-byte pieces per block and row");
    FRAD_DYN_SMEM(smem);
    long long f = blockIdx.x;
    const long long r = f >> 3;
    const int part = (int)(r & 1);
    f = (r >> 1) * 8 + (f & 7);
    if (f >= g.n_frames) return;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<float>* buf = reinterpret_cast<cx<float>*>(smem) + (long long)cf * SLOTS;
    const bool le = g.le != 0, be_in = (g.dtype & 1) != 0;
    const unsigned char* src = pcm + ((frame_base(g, f) * C) << 2) + part * (IW * 4);
    uint32_t w[RPT][IW];
#pragma unroll
    for (int i = 0; i < RPT; ++i) load_words<IW>(src + (long long)(threadIdx.x + i * T) * ROW_IN, w[i]);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int m = makhoul(threadIdx.x + i * T, N);
#pragma unroll
        for (int j = 0; j < CG; ++j) xslot<float, SH>(smem, j, SLOTS, m) = u2f(be_in ? bswap32(w[i][j]) : w[i][j]);
    }
    __syncthreads();
    int tt = t; FRAD_OPAQUE(tt);
    fft_team<float, LOG2M, false>(buf, tt, tw);
    dct_post<float, LOG2M>(buf, tt, post);
    __syncthreads();
    unsigned char* dst = payload + f * g.payload_stride + part * (OW * 4);
    uint32_t mxf = 0;                                         // max |X| as float32 bits: ordered like the values, and a NaN outranks +Inf
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int k = threadIdx.x + i * T;
        uint32_t row[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) row[q] = 0;
#pragma unroll
        for (int j = 0; j < CG; ++j) {
            const float v = xslot<float, SH>(smem, j, SLOTS, k);
            const uint32_t a = f2u(v) & 0x7fffffffu; mxf = a > mxf ? a : mxf;
            const u64 code = storage_code<float>(v, BITS);
            if constexpr (BITS == 16) { uint32_t c = (uint32_t)code & 0xffffu; if (!le) c = bswap16(c); row[j >> 1] |= c << (16 * (j & 1)); }
            else if constexpr (BITS == 32) { row[j] = le ? (uint32_t)code : bswap32((uint32_t)code); }
            else { const u64 c = le ? code : bswap64(code); row[2 * j] = (uint32_t)c; row[2 * j + 1] = (uint32_t)(c >> 32); }
        }
        store_words<OW>(dst + (long long)k * ROW_OUT, row);
    }
    if (absmax != nullptr) {                                  // np.max(np.abs(freqs)): NaN if any coefficient is (profile0.py:24)
        u64 mx = mxf > 0x7f800000u ? 0x7ff8000000000000ULL : d2u((double)u2f(mxf));
        mx = wave_max_u64(mx);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<u64*>(absmax) + f, mx);
    }
}

// 1 = launched; 0 = not this kernel's geometry.  float32 PCM (either byte order), C = 8 channels at N = 4096 or C = 16 at
// N = 2048 ... : C / 2 channels x N / 2 complex float32 points = 64 KiB per block; 16-byte aligned rows both sides; whole frames.
int launch_p0_fwd_half32(int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
                         const Tables& tb, const Geom& g, int ai, int ao) {
    if (tune("FRAD_TUNE_NO_HALF32")) return 0;
    if (lg != 2 || (g.dtype >> 3) != 2 || !ai || !ao || g.n_valid != g.N || g.n_frames > 0x3fffffffLL) return 0;
    if (c.log2m != 11 || g.C != 8 || !(g.bits == 32 || g.bits == 64)) return 0;      // (16-bit rows: 8 bytes per block, not a 16-byte piece)
    const cx<float>* tw = static_cast<const cx<float>*>(tb.tw);
    const cx<float>* post = static_cast<const cx<float>*>(tb.post);
    const long long nb = ((g.n_frames + 7) / 8) * 16;
    const dim3 grid((unsigned)nb), blk(4 * Plan<11>::TEAM);
    const size_t lds = (size_t)4 * 2048 * 8;
#define GO(B) do { allow_lds(k_p0_fwd_half32<11, 4, B>, lds); hipLaunchKernelGGL((k_p0_fwd_half32<11, 4, B>), grid, blk, lds, s, pcm, pay, am, tw, post, g); } while (0)
    if (g.bits == 32) GO(32); else GO(64);
#undef GO
    return 1;
}

}  // namespace frad
