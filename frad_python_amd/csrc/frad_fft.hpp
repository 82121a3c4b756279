// frad_fft.hpp -- LDS-resident Stockham FFT building blocks for gfx950 (CDNA4), wave64.
//
// The reference's Fourier arithmetic is scipy.fft.dct/idct (pocketfft), call sites
// /root/reference/src/libfrad/fourier/profile0.py:21,69 and profile1.py:21,77.  Here an N-point
// DCT-II is computed as an M = N/2 point complex FFT (Makhoul's even/odd permutation packs the
// real sequence into M complex points) plus one pair-wise twiddle step.  A "team" of TEAM lanes
// owns one channel of one frame; every lane keeps P = M/TEAM points in registers per pass, does
// radix-R butterflies on them (R <= 16) and exchanges through LDS between passes.  No MFMA:
// the work is a bandwidth-bound butterfly network, not a dense contraction.
#pragma once
#include "frad_platform.hpp"

namespace frad {

template <typename T> struct cx { T x, y; };
template <> struct __attribute__((aligned(16))) cx<double> { double x, y; };
template <> struct __attribute__((aligned(8))) cx<float> { float x, y; };

template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> w) {
    return {fma(a.x, w.x, -(a.y * w.y)), fma(a.x, w.y, a.y * w.x)};
}
template <typename T> __device__ __forceinline__ cx<T> conj(cx<T> a) { return {a.x, -a.y}; }
// multiply by -i (forward) / +i (inverse)
template <bool INV, typename T> __device__ __forceinline__ cx<T> mul_mi(cx<T> a) {
    if constexpr (INV) return {-a.y, a.x}; else return {a.y, -a.x};
}

// LDS layout: XOR swizzle, no padding.  Slot i lives at i ^ ((i >> SH) & mask), SH = log2 of the
// plan's first radix, mask = 7 for 16-byte slots (ds_write_b128 serves 8 lanes x 16 B per pass over
// 32 banks) and 15 for 8-byte slots.  Lane-contiguous accesses stay inside aligned 16-slot rows
// (conflict-free for every ds_read/ds_write lane group) and the first pass's lane-strided Stockham
// scatter (stride R slots) is spread over all banks instead of hitting one.
// SH == 100 selects the layout of the 4-16-16 inverse plan (frad_persistent.hpp): slot bits 8-9 and bit 6 are
// folded into bits 2-3 / 2, which keeps its three access shapes (lane + 64 j, lane/4 + 16 j + 256 (lane & 3),
// and the 16-strided Stockham scatter of its middle pass) on distinct banks.
template <typename T, int SH> __device__ __forceinline__ int phys(int i) {
    if constexpr (SH < 0) return i;
    else if constexpr (SH == 100) return i ^ ((((i >> 8) & 3) << 2) ^ (((i >> 6) & 1) << 2));
    else return i ^ ((i >> SH) & (sizeof(cx<T>) == 16 ? 7 : 15));
}
// phys(t + c) for c a multiple of 16: the swizzle only touches the low 3-4 bits, which c leaves alone, so
// the constant stays outside the XOR and folds into the ds_read/ds_write immediate offset; the XOR term
// takes two values at most over a pass, i.e. two VALU ops instead of four per access.
template <typename T, int SH> __device__ __forceinline__ int phys_tc(int t, int c) {
    if constexpr (SH < 0) return t + c;
    else if constexpr (SH == 100) return phys<T, SH>(t + c);
    else return c + (t ^ (((t >> SH) + (c >> SH)) & (sizeof(cx<T>) == 16 ? 7 : 15)));
}
__host__ __device__ constexpr int padded_slots(int m) { return m; }

// constants (correctly rounded)
template <typename T> struct K {
    static constexpr T s2 = (T)0.70710678118654752440084436210485L;   // sqrt(1/2)
    static constexpr T c8 = (T)0.92387953251128675612818318939679L;   // cos(pi/8)
    static constexpr T s8 = (T)0.38268343236508977172845998403040L;   // sin(pi/8)
};

// v *= W_16^e (forward) or its conjugate (inverse), e in {0,1,2,3,4,6,9}
template <int E, bool INV, typename T> __device__ __forceinline__ cx<T> mul_w16(cx<T> a) {
    if constexpr (E == 0) return a;
    else if constexpr (E == 4) return mul_mi<INV>(a);
    else if constexpr (E == 2) {            // (1 - i)/sqrt2 forward
        if constexpr (INV) return {(a.x - a.y) * K<T>::s2, (a.x + a.y) * K<T>::s2};
        else return {(a.x + a.y) * K<T>::s2, (a.y - a.x) * K<T>::s2};
    } else if constexpr (E == 6) {          // (-1 - i)/sqrt2 forward
        if constexpr (INV) return {-(a.x + a.y) * K<T>::s2, (a.x - a.y) * K<T>::s2};
        else return {(a.y - a.x) * K<T>::s2, -(a.x + a.y) * K<T>::s2};
    } else {
        constexpr T c = (E == 1) ? K<T>::c8 : (E == 3) ? K<T>::s8 : -K<T>::c8;       // E == 9: -W16^1
        constexpr T s = (E == 1) ? K<T>::s8 : (E == 3) ? K<T>::c8 : -K<T>::s8;
        cx<T> w = {c, INV ? s : -s};
        return cmul(a, w);
    }
}

template <bool INV, typename T> __device__ __forceinline__ void bfly2(cx<T>& a, cx<T>& b) {
    cx<T> t = a - b; a = a + b; b = t;
}
// in-place 4-point DFT, natural order out
template <bool INV, typename T> __device__ __forceinline__ void bfly4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d) {
    cx<T> t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_mi<INV>(b - d);
    a = t0 + t2; c = t0 - t2; b = t1 + t3; d = t1 - t3;
}

// R-point DFT of v[0..R-1], natural order in and out
template <int R, bool INV, typename T> __device__ __forceinline__ void dft(cx<T> (&v)[R]) {
    if constexpr (R == 2) {
        bfly2<INV>(v[0], v[1]);
    } else if constexpr (R == 4) {
        bfly4<INV>(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {
        // n = c + 2a : DFT4 over a for c = 0,1 ; twiddle W8^(c*a') ; DFT2 over c -> X[a' + 4c']
        bfly4<INV>(v[0], v[2], v[4], v[6]);
        bfly4<INV>(v[1], v[3], v[5], v[7]);
        v[3] = mul_w16<2, INV>(v[3]);
        v[5] = mul_w16<4, INV>(v[5]);
        v[7] = mul_w16<6, INV>(v[7]);
        // now u_0[a'] = v[2a'], u_1[a'] = v[2a'+1]
        cx<T> o[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) { o[a] = v[2 * a] + v[2 * a + 1]; o[a + 4] = v[2 * a] - v[2 * a + 1]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = o[i];
    } else {
        static_assert(R == 16, "radix");
        // n = c + 4a : DFT4 over a (c fixed) ; twiddle W16^(c*a') ; DFT4 over c -> X[a' + 4c']
#pragma unroll
        for (int c = 0; c < 4; ++c) bfly4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
        // u_c[a'] = v[c + 4a']
        v[5] = mul_w16<1, INV>(v[5]);  v[9] = mul_w16<2, INV>(v[9]);   v[13] = mul_w16<3, INV>(v[13]);
        v[6] = mul_w16<2, INV>(v[6]);  v[10] = mul_w16<4, INV>(v[10]); v[14] = mul_w16<6, INV>(v[14]);
        v[7] = mul_w16<3, INV>(v[7]);  v[11] = mul_w16<6, INV>(v[11]); v[15] = mul_w16<9, INV>(v[15]);
        // (index c + 4a' holds u_c[a'] * W16^(c a'))
#pragma unroll
        for (int a = 0; a < 4; ++a) bfly4<INV>(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]);
        // v[4a' + c'] = X[a' + 4c'] -> transpose to natural order
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = a + 1; c < 4; ++c) { cx<T> t = v[4 * a + c]; v[4 * a + c] = v[4 * c + a]; v[4 * c + a] = t; }
    }
}

// team-level synchronisation: lanes of one wave run in lockstep and the LDS serves a wave's
// instructions in order, so a team that fits in a wave only needs the compiler kept honest.
template <int TEAM, bool LDSONLY = false> __device__ __forceinline__ void team_sync() {
    if constexpr (TEAM <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if constexpr (LDSONLY) {
        FRAD_LDS_BARRIER();                      // pipelined kernels: do not drain global loads/stores
    } else {
        __syncthreads();
    }
}

// One Stockham pass of radix R with accumulated stride NS over the M-point array `buf`
// (padded layout).  `t` = lane index inside the team.  `tw` = W_M^k table, k in [0, M).
template <typename T, int M, int TEAM, int R, int NS, bool INV, int SH>
__device__ __forceinline__ void fft_pass(cx<T>* buf, int t, const cx<T>* __restrict__ tw) {
    constexpr int NB = M / R / TEAM;            // butterflies per lane
    static_assert(NB >= 1 && NB * R * TEAM == M, "pass geometry");
    cx<T> v[NB][R];
    cx<T> w[NB][R];
    __builtin_amdgcn_sched_barrier(0);          // keep this pass's loads out of the previous pass's registers
    // twiddles first: they come from L2 and depend on the lane only, so their latency overlaps the
    // LDS reads below instead of following them
    if constexpr (NS > 1) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int k = (t + nb * TEAM) & (NS - 1);
#pragma unroll
            for (int j = 1; j < R; ++j) w[nb][j] = tw[j * k * (M / (NS * R))];
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = t + nb * TEAM;
#pragma unroll
        for (int j = 0; j < R; ++j) v[nb][j] = buf[phys<T, SH>(b + j * (M / R))];
    }
    team_sync<TEAM>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = t + nb * TEAM;
        const int k = b & (NS - 1);
        if constexpr (NS > 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) {
                cx<T> ww = w[nb][j];
                if constexpr (INV) ww.y = -ww.y;
                v[nb][j] = cmul(v[nb][j], ww);
            }
        }
        dft<R, INV>(v[nb]);
        const int base = (b - k) * R + k;
#pragma unroll
        for (int j = 0; j < R; ++j) buf[phys<T, SH>(base + j * NS)] = v[nb][j];
    }
    team_sync<TEAM>();
}

// The same pass with its twiddles in a lane-linear table (the persistent kernels keep it in LDS):
// entry ((nb_eff * (R-1) + j-1) * KW + kk) = W_{NS*R}^(j * k), with KW = min(NS, TEAM), kk = t mod KW,
// nb_eff = nb when NS > TEAM else 0 -- consecutive lanes read consecutive (or identical, broadcast)
// slots, so the reads are free of bank conflicts, and they are issued next to their use.
template <int TEAM, int R, int NS, int M> __host__ __device__ constexpr int pass_table_size() {
    return NS == 1 ? 0 : (NS > TEAM ? (M / R / TEAM) : 1) * (R - 1) * (NS > TEAM ? TEAM : NS);
}
template <typename T, int M, int TEAM, int R, int NS, bool INV, int SH>
__device__ __forceinline__ void fft_pass_lt(cx<T>* buf, int t, const cx<T>* ptab) {
    constexpr int NB = M / R / TEAM, KW = NS > TEAM ? TEAM : NS;
    static_assert(NB >= 1 && NB * R * TEAM == M, "pass geometry");
    cx<T> v[NB][R];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if constexpr (TEAM % 16 == 0 && (M / R) % 16 == 0) v[nb][j] = buf[phys_tc<T, SH>(t, nb * TEAM + j * (M / R))];
            else v[nb][j] = buf[phys<T, SH>(t + nb * TEAM + j * (M / R))];
        }
    }
    team_sync<TEAM, true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = t + nb * TEAM;
        const int k = b & (NS - 1);
        if constexpr (NS > 1) {
            const int kk = t & (KW - 1), nbe = NS > TEAM ? nb : 0;
#pragma unroll
            for (int j = 1; j < R; ++j) {
                cx<T> ww = ptab[(nbe * (R - 1) + (j - 1)) * KW + kk];
                if constexpr (INV) ww.y = -ww.y;
                v[nb][j] = cmul(v[nb][j], ww);
            }
        }
        dft<R, INV>(v[nb]);
        const int base = (b - k) * R + k;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if constexpr (NS * R == M && TEAM % 16 == 0 && NS % 16 == 0) buf[phys_tc<T, SH>(t, nb * TEAM + j * NS)] = v[nb][j];   // last pass: b + j*NS
            else buf[phys<T, SH>(base + j * NS)] = v[nb][j];
        }
    }
    team_sync<TEAM, true>();
}

// Pass schedules.  P = points per lane = max radix, TEAM = M / P lanes per channel-frame.
template <int LOG2M> struct Plan;
template <> struct Plan<6>  { static constexpr int TEAM = 16,  SH = 2; };   // M = 64   : 4 4 4
template <> struct Plan<7>  { static constexpr int TEAM = 32,  SH = 2; };   // M = 128  : 4 4 4 2
template <> struct Plan<8>  { static constexpr int TEAM = 64,  SH = 2; };   // M = 256  : 4 4 4 4
template <> struct Plan<9>  { static constexpr int TEAM = 64,  SH = 3; };   // M = 512  : 8 8 8
template <> struct Plan<10> { static constexpr int TEAM = 64,  SH = 4; };   // M = 1024 : 16 16 4
template <> struct Plan<11> { static constexpr int TEAM = 128, SH = 4; };   // M = 2048 : 16 16 8
template <> struct Plan<12> { static constexpr int TEAM = 256, SH = 4; };   // M = 4096 : 16 16 16
template <> struct Plan<13> { static constexpr int TEAM = 512, SH = 4; };   // M = 8192 : 16 16 16 2

template <typename T, int LOG2M, bool INV>
__device__ __forceinline__ void fft_team(cx<T>* buf, int t, const cx<T>* __restrict__ tw) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SH = Plan<LOG2M>::SH;
    if constexpr (LOG2M == 6) {
        fft_pass<T, M, TEAM, 4, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 4, 4, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 4, 16, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 7) {
        fft_pass<T, M, TEAM, 4, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 4, 4, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 4, 16, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 2, 64, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 8) {
        fft_pass<T, M, TEAM, 4, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 4, 4, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 4, 16, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 4, 64, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 9) {
        fft_pass<T, M, TEAM, 8, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 8, 8, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 8, 64, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 10) {
        fft_pass<T, M, TEAM, 16, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 16, 16, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 4, 256, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 11) {
        fft_pass<T, M, TEAM, 16, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 16, 16, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 8, 256, INV, SH>(buf, t, tw);
    } else if constexpr (LOG2M == 12) {
        fft_pass<T, M, TEAM, 16, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 16, 16, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 16, 256, INV, SH>(buf, t, tw);
    } else {
        static_assert(LOG2M == 13, "plan");
        fft_pass<T, M, TEAM, 16, 1, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 16, 16, INV, SH>(buf, t, tw);
        fft_pass<T, M, TEAM, 16, 256, INV, SH>(buf, t, tw); fft_pass<T, M, TEAM, 2, 4096, INV, SH>(buf, t, tw);
    }
}

// ---------------------------------------------------------------------------------------------
// DCT-II <-> packed-FFT glue.  Real slot r of a channel buffer lives in complex slot r>>1,
// component r&1 (same padded layout), so the N reals X[k] overlay the M complex points.
// ---------------------------------------------------------------------------------------------
template <typename T, int SH> __device__ __forceinline__ T& real_slot(cx<T>* buf, int r) {
    return reinterpret_cast<T*>(buf + phys<T, SH>(r >> 1))[r & 1];
}
// position of time sample n in Makhoul's permuted sequence v (v[i] = x[2i], v[N-1-i] = x[2i+1])
__device__ __forceinline__ int makhoul(int n, int N) { return (n & 1) ? N - 1 - (n >> 1) : (n >> 1); }

// post[k] = { w_k, g_k } with w_k = exp(-i pi k / 2N), g_k = -i w_k exp(-2 pi i k / N), k in [0, M/2]
//
// forward: Z (FFT of the packed sequence) -> X[k] = (1/N) sum x[n] cos(pi k (2n+1) / 2N), in place.
template <typename T, int LOG2M, int PS = 2, int TEAM = Plan<LOG2M>::TEAM, int SH = Plan<LOG2M>::SH, bool LDSONLY = false>
__device__ __forceinline__ void dct_post(cx<T>* buf, int t, const cx<T>* __restrict__ post, T extra = (T)1) {
    constexpr int GOFF = PS == 2 ? 1 : (1 << LOG2M) / 2 + 1;   // where g_k lives relative to w_k
    constexpr int M = 1 << LOG2M, N = 2 * M;
    constexpr int PP = (M / 2) / TEAM;         // pairs per lane (pair M/2 goes to lane 0 on top)
    // `extra` is an exact power of two (a deferred PCM normalisation): scaling by it commutes with rounding
    const T sc = ((T)1 / (T)(2 * N)) * extra;
    const T sc2 = (K<T>::s2 / (T)(2 * N)) * extra;
    cx<T> S[PP + 1], D[PP + 1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i <= PP; ++i) {
        const int k = (i < PP) ? t + i * TEAM : M / 2;
        if (i == PP && t != 0) continue;
        const cx<T> zk = (i < PP && TEAM % 16 == 0) ? buf[phys_tc<T, SH>(t, i * TEAM)] : buf[phys<T, SH>(k)];
        const cx<T> zp = conj(buf[phys<T, SH>((M - k) & (M - 1))]);
        const cx<T> p = cmul(zk + zp, post[PS * k]), q = cmul(zk - zp, post[PS * k + GOFF]);
        S[i] = p + q; D[i] = p - q;
    }
    team_sync<TEAM, LDSONLY>();
#pragma unroll
    for (int i = 0; i <= PP; ++i) {
        const int k = (i < PP) ? t + i * TEAM : M / 2;
        if (i == PP && t != 0) continue;
        real_slot<T, SH>(buf, k) = S[i].x * sc;
        if (k > 0) real_slot<T, SH>(buf, N - k) = -S[i].y * sc;
        if (k < M / 2) {
            real_slot<T, SH>(buf, M - k) = (D[i].x - D[i].y) * sc2;
            if (k > 0) real_slot<T, SH>(buf, M + k) = (D[i].x + D[i].y) * sc2;
        }
    }
    team_sync<TEAM, LDSONLY>();
}

// inverse: X (N reals, 'forward'-normalised DCT-II coefficients) -> Z' = Z / M, in place, so that
// the unscaled inverse FFT returns the packed time sequence.
template <typename T, int LOG2M, int PS = 2, int TEAM = Plan<LOG2M>::TEAM, int SH = Plan<LOG2M>::SH, bool LDSONLY = false>
__device__ __forceinline__ void dct_pre_inverse(cx<T>* buf, int t, const cx<T>* __restrict__ post) {
    constexpr int GOFF = PS == 2 ? 1 : (1 << LOG2M) / 2 + 1;
    constexpr int M = 1 << LOG2M, N = 2 * M;
    constexpr int PP = (M / 2) / TEAM;
    cx<T> A[PP + 1], B[PP + 1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i <= PP; ++i) {
        const int k = (i < PP) ? t + i * TEAM : M / 2;
        if (i == PP && t != 0) continue;
        const T xk = real_slot<T, SH>(buf, k);
        const T xnk = k > 0 ? real_slot<T, SH>(buf, N - k) : (T)0;
        const T a = real_slot<T, SH>(buf, M - k), b = real_slot<T, SH>(buf, k > 0 ? M + k : M);
        const cx<T> u = {xk, -xnk};
        const cx<T> s = {(a + b) * K<T>::s2, (b - a) * K<T>::s2};
        A[i] = cmul(u + s, conj(post[PS * k]));
        B[i] = cmul(u - s, conj(post[PS * k + GOFF]));
    }
    team_sync<TEAM, LDSONLY>();
#pragma unroll
    for (int i = 0; i <= PP; ++i) {
        const int k = (i < PP) ? t + i * TEAM : M / 2;
        if (i == PP && t != 0) continue;
        buf[phys<T, SH>(k)] = A[i] + B[i];
        if (k > 0 && k < M / 2) buf[phys<T, SH>(M - k)] = conj(A[i] - B[i]);
    }
    team_sync<TEAM, LDSONLY>();
}

}  // namespace frad
