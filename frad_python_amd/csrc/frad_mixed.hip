// frad_mixed.hip -- O(N log N) DCT for frame lengths N = 2 r 2^p, r in {3, 5, 7}.
//
// Who needs them: profile 1's compact frame sizes {160, 192, 224} x 2^n (fourier/profiles.py:14-23 -- 24 of its 32 legal
// sizes) and a clip's last lossless frame (encoder.py:72-93: BASELINE config 3's 896-sample tails = 7 x 128).  Round 2
// ran them as dense cosine products (profile 1) or Bluestein convolutions (profile 0, two FFTs of twice the length).
//
// Same route as the power-of-two kernels -- Makhoul's even/odd permutation, the N reals packed as M = N/2 complex points,
// an M-point complex FFT, one pair step (frad_fft.hpp) -- with the FFT split as M = r P, P = 2^p:
//     z_b[a] = z[r a + b]                               r decimated sequences (LDS sub-buffers of P slots)
//     Y_b    = FFT_P(z_b)                               the LDS-resident power-of-two FFT (fft_team), one team per sub-buffer
//     Z[k1 + P k2] = sum_b (W_M^(b k1) Y_b[k1]) W_r^(b k2)      twiddle + one radix-r butterfly per k1, in place
// and backwards with conjugate roots for the inverse.  One frame per block; LDS holds the frame twice (time / coefficient
// plane [C][N] float64 and the complex plane [C][M]), the geometry of the direct kernels these replace, whose stage-in,
// pack / unpack, store and profile-1 quantiser / dequantiser run unchanged around the transform.  float64 throughout
// (float32 / float16 PCM is widened exactly: more accurate than the reference's float32 transform, inside its tolerance).
#include "frad_p1.hpp"
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

namespace frad {
namespace {

struct MixedTab {
    const cx<double>* tw2;      // [(b - 1) * P + k1] = W_M^(b k1), b = 1 .. r-1
    const cx<double>* post;     // [2 k] = w_k, [2 k + 1] = g_k, k = 0 .. M/2 (frad_fft.hpp dct_post)
    const cx<double>* twp;      // W_P^k, the power-of-two FFT's table
    int r, log2p;
};

// cos / sin (2 pi m / R), m = 0 .. R-1, correctly rounded (constexpr functions: usable with the unrolled loops' indices)
__host__ __device__ constexpr double rk_cos(int R, int m) {
    return m == 0 ? 1.0
         : R == 3 ? -0.5
         : R == 5 ? ((m == 1 || m == 4) ? 0.30901699437494742410229341718282 : -0.80901699437494742410229341718282)
         : ((m == 1 || m == 6) ? 0.62348980185873353052500488400424 : (m == 2 || m == 5) ? -0.22252093395631440428890256449679
                                                                                          : -0.90096886790241912623610231950745);
}
__host__ __device__ constexpr double rk_sin(int R, int m) {
    const int a = m <= R / 2 ? m : R - m;                      // sin(2 pi (R - m) / R) = -sin(2 pi m / R)
    const double v = a == 0 ? 0.0
         : R == 3 ? 0.86602540378443864676372317075294
         : R == 5 ? (a == 1 ? 0.95105651629515357211643933337938 : 0.58778525229247312916870595463907)
         : (a == 1 ? 0.78183148246802980870844452667406 : a == 2 ? 0.97492791218182360701813168299393 : 0.43388373911755812047576833284836);
    return m <= R / 2 ? v : -v;
}

// R-point DFT in registers, e^(-2 pi i j k / R) (forward) or its conjugate: the sums / differences of the mirrored inputs
// halve the products (y_j e^(-i t) + y_(R-j) e^(+i t) = (y_j + y_(R-j)) cos t - i (y_j - y_(R-j)) sin t)
template <int R, bool INV>
__device__ __forceinline__ void dft_small(cx<double> (&y)[R]) {
    constexpr int H = (R - 1) / 2;
    cx<double> sm[H], df[H], out[R];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = y[j] + y[R - j]; df[j - 1] = y[j] - y[R - j]; }
    out[0] = y[0];
#pragma unroll
    for (int j = 0; j < H; ++j) out[0] = out[0] + sm[j];
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        cx<double> A = y[0], B = {0.0, 0.0};
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const int m = (j * k) % R;
            const double cc = rk_cos(R, m), ss = rk_sin(R, m);
            A.x = fma(sm[j - 1].x, cc, A.x); A.y = fma(sm[j - 1].y, cc, A.y);
            B.x = fma(df[j - 1].x, ss, B.x); B.y = fma(df[j - 1].y, ss, B.y);
        }
        const cx<double> lo = {A.x + B.y, A.y - B.x}, hi = {A.x - B.y, A.y + B.x};      // A - i B, A + i B
        out[k] = INV ? hi : lo; out[R - k] = INV ? lo : hi;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) y[k] = out[k];
}

// slot of point i inside a P-slot sub-buffer: the swizzle of the power-of-two plan (frad_fft.hpp phys<double, SH>)
__device__ __forceinline__ int pslot_p(int i, int sh) { return i ^ ((i >> sh) & 7); }
__device__ __forceinline__ int plan_sh(int log2p) { return log2p <= 8 ? 2 : log2p == 9 ? 3 : 4; }

template <bool INV>
__device__ __forceinline__ void sub_ffts(cx<double>* zb, int nsub, int log2p, const cx<double>* __restrict__ twp) {
    auto run = [&](auto tag) {
        constexpr int L2 = decltype(tag)::value, TEAM = Plan<L2>::TEAM;
        const int team = threadIdx.x / TEAM, nteams = blockDim.x / TEAM;
        int t = threadIdx.x - team * TEAM;
        // A wave holds TPW teams; when the last round leaves some of them without a sub-buffer they repeat the wave's first
        // one in lockstep (same reads, same writes, the passes' wave-level barriers in between): every lane of an active wave
        // runs the same number of passes, and a wave without any work skips the round altogether.
        constexpr int TPW = 64 / TEAM;
        for (int s = team; s - (team % TPW) < nsub; s += nteams) {
            const int se = s < nsub ? s : s - (team % TPW);
            FRAD_OPAQUE(t);
            fft_team<double, L2, INV>(zb + ((long long)se << L2), t, twp);
        }
    };
    switch (log2p) {
        case 6: run(std::integral_constant<int, 6>{}); break;
        case 7: run(std::integral_constant<int, 7>{}); break;
        case 8: run(std::integral_constant<int, 8>{}); break;
        case 9: run(std::integral_constant<int, 9>{}); break;
        default: run(std::integral_constant<int, 10>{}); break;
    }
}

// twiddle + radix-r butterfly over the r sub-buffers of every channel, in place: (buffer b, slot k1) -> (buffer k2, slot k1).
// (R is a template parameter: a register array indexed under a run-time radix would live in scratch memory.)
template <int R, bool INV>
__device__ __forceinline__ void radix_pass_r(cx<double>* zb, int C, const MixedTab& mt) {
    const int log2p = mt.log2p, P = 1 << log2p, sh = plan_sh(log2p);
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) {
        const int c = i >> log2p, k1 = i & (P - 1);
        cx<double>* base = zb + ((long long)(c * R) << log2p) + pslot_p(k1, sh);
        cx<double> y[R];
#pragma unroll
        for (int b = 0; b < R; ++b) y[b] = base[(long long)b << log2p];
#pragma unroll
        for (int b = 1; b < R; ++b) {
            const cx<double> w = mt.tw2[(b - 1) * P + k1];
            y[b] = cmul(y[b], INV ? conj(w) : w);
        }
        dft_small<R, INV>(y);
#pragma unroll
        for (int b = 0; b < R; ++b) base[(long long)b << log2p] = y[b];
    }
}
template <bool INV>
__device__ __forceinline__ void radix_pass(cx<double>* zb, int C, const MixedTab& mt) {
    if (mt.r == 3) radix_pass_r<3, INV>(zb, C, mt); else if (mt.r == 5) radix_pass_r<5, INV>(zb, C, mt); else radix_pass_r<7, INV>(zb, C, mt);
}

// time plane xa [C][N] (natural order) -> coefficient plane xa [C][N]; zb = complex plane [C][M]
// (`permuted`: the time plane is in Makhoul's order v -- what the channel-group stage-in / store helpers use -- where the packed
//  points are simply z[m] = (v[2m], v[2m + 1]))
__device__ __forceinline__ void mixed_forward(double* xa, cx<double>* zb, int N, int C, const MixedTab& mt, bool permuted = false) {
    const int r = mt.r, log2p = mt.log2p, P = 1 << log2p, M = N / 2, H = M / 2, sh = plan_sh(log2p);
    auto zslot = [&](int c, int m) -> cx<double>& {           // decimated layout: buffer m mod r, point m / r
        const int a = m / r, b = m - a * r;
        return zb[((long long)(c * r + b) << log2p) + pslot_p(a, sh)];
    };
    for (int i = threadIdx.x; i < C * H; i += blockDim.x) {   // Makhoul's permutation, two packed points per quad of samples
        const int c = i / H, q = i - c * H;
        if (permuted) {
            const double* v = xa + (long long)c * N;
            zslot(c, q) = cx<double>{v[2 * q], v[2 * q + 1]};
            zslot(c, M - 1 - q) = cx<double>{v[N - 2 - 2 * q], v[N - 1 - 2 * q]};
        } else {
            const double* x = xa + (long long)c * N + 4 * q;
            zslot(c, q) = cx<double>{x[0], x[2]};
            zslot(c, M - 1 - q) = cx<double>{x[3], x[1]};
        }
    }
    __syncthreads();
    sub_ffts<false>(zb, C * r, log2p, mt.twp);
    __syncthreads();
    radix_pass<false>(zb, C, mt);
    __syncthreads();
    // pair step (frad_fft.hpp dct_post): Z[k], Z[M - k] -> X[k], X[N - k], X[M - k], X[M + k]; Z[k] sits in buffer k / P
    const double sc = 1.0 / (double)(2 * N), sc2 = K<double>::s2 / (double)(2 * N);
    auto zat = [&](int c, int k) -> cx<double> { return zb[((long long)(c * r + (k >> log2p)) << log2p) + pslot_p(k & (P - 1), sh)]; };
    for (int i = threadIdx.x; i < C * (H + 1); i += blockDim.x) {
        const int c = i / (H + 1), k = i - c * (H + 1);
        const cx<double> zk = zat(c, k), zp = conj(zat(c, k == 0 ? 0 : M - k));
        const cx<double> p = cmul(zk + zp, mt.post[2 * k]), q = cmul(zk - zp, mt.post[2 * k + 1]);
        const cx<double> S = p + q, D = p - q;
        double* X = xa + (long long)c * N;
        X[k] = S.x * sc;
        if (k > 0) X[N - k] = -S.y * sc;
        if (k < H) {
            X[M - k] = (D.x - D.y) * sc2;
            if (k > 0) X[M + k] = (D.x + D.y) * sc2;
        }
    }
    __syncthreads();
}

// coefficient plane xa [C][N] -> time plane xa [C][N] (natural order)
__device__ __forceinline__ void mixed_inverse(double* xa, cx<double>* zb, int N, int C, const MixedTab& mt, bool permuted = false) {
    const int r = mt.r, log2p = mt.log2p, P = 1 << log2p, M = N / 2, H = M / 2, sh = plan_sh(log2p);
    auto zslot = [&](int c, int m) -> cx<double>& {
        const int a = m / r, b = m - a * r;
        return zb[((long long)(c * r + b) << log2p) + pslot_p(a, sh)];
    };
    for (int i = threadIdx.x; i < C * (H + 1); i += blockDim.x) {     // frad_fft.hpp dct_pre_inverse
        const int c = i / (H + 1), k = i - c * (H + 1);
        const double* X = xa + (long long)c * N;
        const double xk = X[k], xnk = k > 0 ? X[N - k] : 0.0;
        const double a = X[M - k], b = X[k > 0 ? M + k : M];
        const cx<double> u = {xk, -xnk};
        const cx<double> s = {(a + b) * K<double>::s2, (b - a) * K<double>::s2};
        const cx<double> A = cmul(u + s, conj(mt.post[2 * k])), B = cmul(u - s, conj(mt.post[2 * k + 1]));
        zslot(c, k) = A + B;
        if (k > 0 && k < H) zslot(c, M - k) = conj(A - B);
    }
    __syncthreads();
    sub_ffts<true>(zb, C * r, log2p, mt.twp);
    __syncthreads();
    radix_pass<true>(zb, C, mt);
    __syncthreads();
    auto zat = [&](int c, int m) -> cx<double> { return zb[((long long)(c * r + (m >> log2p)) << log2p) + pslot_p(m & (P - 1), sh)]; };
    for (int i = threadIdx.x; i < C * H; i += blockDim.x) {
        const int c = i / H, q = i - c * H;
        const cx<double> za = zat(c, q), zc = zat(c, M - 1 - q);
        if (permuted) {
            double* v = xa + (long long)c * N;
            v[2 * q] = za.x; v[2 * q + 1] = za.y; v[N - 2 - 2 * q] = zc.x; v[N - 1 - 2 * q] = zc.y;
        } else {
            double* x = xa + (long long)c * N + 4 * q;
            x[0] = za.x; x[2] = za.y; x[3] = zc.x; x[1] = zc.y;
        }
    }
    __syncthreads();
}

// ---- kernels: one frame per block, 256 threads; LDS = [C][N] float64 | [C][M] complex | (profile 1: quantiser scratch) ----
template <int LG>
__global__ void __launch_bounds__(256) k_p0_fwd_mixed(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                      double* absmax, Geom g, MixedTab mt, int aligned_in, int aligned_out) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* xa = reinterpret_cast<double*>(smem);
    cx<double>* zb = reinterpret_cast<cx<double>*>(smem + (long long)N * C * 8);
    stage_in_pcm<double, LG, -1, false>(pcm, 0, g, f0, 1, N, aligned_in != 0);
    __syncthreads();
    mixed_forward(xa, zb, N, C, mt);
    pack_out_any<double, -1>(0, payload, absmax, g, f0, 1, N, aligned_out != 0);
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p0_inv_mixed(const unsigned char* __restrict__ payload, double* __restrict__ out, Geom g,
                                                      MixedTab mt, int aligned_in) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* xa = reinterpret_cast<double*>(smem);
    cx<double>* zb = reinterpret_cast<cx<double>*>(smem + (long long)N * C * 8);
    unpack_in_any<-1>(payload, 0, g, f0, 1, N, aligned_in != 0);
    __syncthreads();
    mixed_inverse(xa, zb, N, C, mt);
    store_pcm_f64<-1, false>(0, out, g, f0, 1, N);
}

template <int LG>
__global__ void __launch_bounds__(256) k_p1_fwd_mixed(const unsigned char* __restrict__ pcm, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                                      Geom g, P1Tables tb, MixedTab mt, int aligned_in) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* xa = reinterpret_cast<double*>(smem);
    const int cg = g.cg;                                      // channels per pass (< C: the frame's channels do not fit the LDS together)
    cx<double>* zb = reinterpret_cast<cx<double>*>(smem + (long long)N * cg * 8);
    p1_tables_to_lds(smem + 2 * N * cg * 8, cg, tb, N);
    if (cg == C) {
        stage_in_pcm<double, LG, -1, false>(pcm, 0, g, f0, 1, N, aligned_in != 0);
        __syncthreads();
        mixed_forward(xa, zb, N, C, mt);
        p1_quantise<-1>(0, 2 * N * C * 8, N, tb.scale, tb.loss, tb.nb_used, g, f0, 1, q, tq, 0, C, C, tb.f32);
        return;
    }
    for (int c0 = 0; c0 < C; c0 += cg) {                      // the quantiser works channel by channel: groups are independent
        const int cgn = C - c0 < cg ? C - c0 : cg;
        stage_in_pcm_group<double, LG, -1>(pcm, 0, g, f0, N, c0, cgn);
        __syncthreads();
        mixed_forward(xa, zb, N, cgn, mt, true);
        p1_quantise<-1>(0, 2 * N * cg * 8, N, tb.scale, tb.loss, tb.nb_used, g, f0, 1, q, tq, c0, cgn, cg, tb.f32);
        __syncthreads();
    }
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p1_inv_mixed(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
                                                      Geom g, P1Tables tb, MixedTab mt) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* xa = reinterpret_cast<double*>(smem);
    const int cg = g.cg;
    cx<double>* zb = reinterpret_cast<cx<double>*>(smem + (long long)N * cg * 8);
    p1_tables_to_lds(smem + 2 * N * cg * 8, cg, tb, N);
    if (cg == C) {
        p1_dequantise<-1>(0, 2 * N * C * 8, N, tb.scale, g, f0, 1, q, tq, 0, C, C);
        __syncthreads();
        mixed_inverse(xa, zb, N, C, mt);
        store_pcm_f64<-1, false>(0, out, g, f0, 1, N);
        return;
    }
    for (int c0 = 0; c0 < C; c0 += cg) {
        const int cgn = C - c0 < cg ? C - c0 : cg;
        p1_dequantise<-1>(0, 2 * N * cg * 8, N, tb.scale, g, f0, 1, q, tq, c0, cgn, cg);
        __syncthreads();
        mixed_inverse(xa, zb, N, cgn, mt, true);
        store_pcm_group<-1>(0, out, g, f0, N, c0, cgn);
        __syncthreads();
    }
}

// ---- frames wider than a CU: the same decomposition through an HBM workspace (frad_global.hip's rows) -------------------
// zw: complex [rows][M], a row = r sub-buffers of P points (decimated layout on the way in, blocked -- Z[k] in buffer k / P --
// after the radix pass).  Three launches per direction: pack (or the inverse pair step), the P-point sub-FFTs (one block per
// sub-buffer, LDS-resident), radix pass fused with the pair step (or with the un-permutation).
__global__ void __launch_bounds__(256) k_gm_pack(const double* __restrict__ xin, cx<double>* __restrict__ zw, int N, int r, int log2p) {
    const int M = N / 2, H = M / 2;
    const long long row = blockIdx.y;
    const double* x = xin + row * (long long)N;
    cx<double>* z = zw + row * (long long)M;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < H; q += gridDim.x * blockDim.x) {
        const int m0 = q, m1 = M - 1 - q;
        z[((long long)(m0 % r) << log2p) + m0 / r] = cx<double>{x[4 * q], x[4 * q + 2]};
        z[((long long)(m1 % r) << log2p) + m1 / r] = cx<double>{x[4 * q + 3], x[4 * q + 1]};
    }
}

template <int L2, bool INV>
__global__ void __launch_bounds__(Plan<L2>::TEAM) k_gm_sub(cx<double>* __restrict__ zw, const cx<double>* __restrict__ twp) {
    constexpr int P = 1 << L2, TEAM = Plan<L2>::TEAM, SH = Plan<L2>::SH;
    FRAD_DYN_SMEM(smem);
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    cx<double>* g = zw + ((long long)blockIdx.x << L2);
    for (int i = threadIdx.x; i < P; i += TEAM) buf[phys<double, SH>(i)] = g[i];
    __syncthreads();
    int t = threadIdx.x;
    fft_team<double, L2, INV>(buf, t, twp);
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += TEAM) g[i] = buf[phys<double, SH>(i)];
}

// forward: columns k1 and P - k1 of a row -> twiddle + radix-r -> the r pairs (k, M - k) they hold -> pair step -> X
template <int R>
__global__ void __launch_bounds__(256) k_gm_radix_post(const cx<double>* __restrict__ zw, double* __restrict__ out, MixedTab mt, int N, int C,
                                                       long long fstride, long long cstride, long long ostride) {
    const int log2p = mt.log2p, P = 1 << log2p, M = N / 2, H = M / 2;
    const long long row = blockIdx.y;
    const cx<double>* z = zw + row * (long long)M;
    double* o = out + (row / C) * fstride + (row % C) * cstride;
    const double sc = 1.0 / (double)(2 * N), sc2 = K<double>::s2 / (double)(2 * N);
    for (int k1 = blockIdx.x * blockDim.x + threadIdx.x; k1 <= P / 2; k1 += gridDim.x * blockDim.x) {
        const int k1m = (P - k1) & (P - 1);
        cx<double> ya[R], yb[R];
#pragma unroll
        for (int b = 0; b < R; ++b) {
            ya[b] = z[((long long)b << log2p) + k1]; yb[b] = z[((long long)b << log2p) + k1m];
            if (b > 0) { ya[b] = cmul(ya[b], mt.tw2[(b - 1) * P + k1]); yb[b] = cmul(yb[b], mt.tw2[(b - 1) * P + k1m]); }
        }
        dft_small<R, false>(ya); dft_small<R, false>(yb);
        // ya[k2] = Z[k1 + P k2], yb[k2] = Z[k1m + P k2]; partner of k = k1 + P k2 is M - k = k1m + P (r - 1 - k2) (k1 > 0) or P (r - k2) (k1 = 0)
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2) {
            const int k = k1 + (k2 << log2p);
            if (k > H) continue;                              // each pair once: k in [0, M/2]
            cx<double> zm;
            if (k1 == 0) zm = k2 == 0 ? ya[0] : ya[(R - k2) % R]; else zm = yb[R - 1 - k2];
            const cx<double> zk = ya[k2], zp = conj(zm);
            const cx<double> p = cmul(zk + zp, mt.post[2 * k]), q = cmul(zk - zp, mt.post[2 * k + 1]);
            const cx<double> S = p + q, D = p - q;
            o[(long long)k * ostride] = S.x * sc;
            if (k > 0) o[(long long)(N - k) * ostride] = -S.y * sc;
            if (k < H) {
                o[(long long)(M - k) * ostride] = (D.x - D.y) * sc2;
                if (k > 0) o[(long long)(M + k) * ostride] = (D.x + D.y) * sc2;
            }
        }
        // the pairs whose smaller member sits in column k1m (k = k1m + P k2 <= M/2), unless the two columns coincide
        if (k1m != k1) {
#pragma unroll
            for (int k2 = 0; k2 < R; ++k2) {
                const int k = k1m + (k2 << log2p);
                if (k > H) continue;
                const cx<double> zk = yb[k2], zp = conj(ya[R - 1 - k2]);
                const cx<double> p = cmul(zk + zp, mt.post[2 * k]), q = cmul(zk - zp, mt.post[2 * k + 1]);
                const cx<double> S = p + q, D = p - q;
                o[(long long)k * ostride] = S.x * sc;
                o[(long long)(N - k) * ostride] = -S.y * sc;
                if (k < H) {
                    o[(long long)(M - k) * ostride] = (D.x - D.y) * sc2;
                    o[(long long)(M + k) * ostride] = (D.x + D.y) * sc2;
                }
            }
        }
    }
}

// inverse, step 1: X row (planar) -> Z' in the decimated layout (frad_fft.hpp dct_pre_inverse)
__global__ void __launch_bounds__(256) k_gm_pre_inverse(const double* __restrict__ xin, cx<double>* __restrict__ zw, MixedTab mt, int N) {
    const int r = mt.r, log2p = mt.log2p, M = N / 2, H = M / 2;
    const long long row = blockIdx.y;
    const double* X = xin + row * (long long)N;
    cx<double>* z = zw + row * (long long)M;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= H; k += gridDim.x * blockDim.x) {
        const double xk = X[k], xnk = k > 0 ? X[N - k] : 0.0;
        const double a = X[M - k], b = X[k > 0 ? M + k : M];
        const cx<double> u = {xk, -xnk};
        const cx<double> s = {(a + b) * K<double>::s2, (b - a) * K<double>::s2};
        const cx<double> A = cmul(u + s, conj(mt.post[2 * k])), B = cmul(u - s, conj(mt.post[2 * k + 1]));
        z[((long long)(k % r) << log2p) + k / r] = A + B;
        if (k > 0 && k < H) { const int m = M - k; z[((long long)(m % r) << log2p) + m / r] = conj(A - B); }
    }
}

// inverse, step 3: conjugate twiddle + radix-r -> z[m1 + P m2] -> Makhoul's permutation undone -> out (strided)
template <int R>
__global__ void __launch_bounds__(256) k_gm_radix_unpack(const cx<double>* __restrict__ zw, double* __restrict__ out, MixedTab mt, int N, int C,
                                                         long long fstride, long long cstride, long long ostride) {
    const int log2p = mt.log2p, P = 1 << log2p, M = N / 2, H = M / 2;
    const long long row = blockIdx.y;
    const cx<double>* z = zw + row * (long long)M;
    double* o = out + (row / C) * fstride + (row % C) * cstride;
    for (int k1 = blockIdx.x * blockDim.x + threadIdx.x; k1 < P; k1 += gridDim.x * blockDim.x) {
        cx<double> y[R];
#pragma unroll
        for (int b = 0; b < R; ++b) {
            y[b] = z[((long long)b << log2p) + k1];
            if (b > 0) y[b] = cmul(y[b], conj(mt.tw2[(b - 1) * P + k1]));
        }
        dft_small<R, true>(y);
#pragma unroll
        for (int m2 = 0; m2 < R; ++m2) {
            const int m = k1 + (m2 << log2p);
            if (m < H) { o[(long long)(4 * m) * ostride] = y[m2].x; o[(long long)(4 * m + 2) * ostride] = y[m2].y; }
            else { const int q = M - 1 - m; o[(long long)(4 * q + 3) * ostride] = y[m2].x; o[(long long)(4 * q + 1) * ostride] = y[m2].y; }
        }
    }
}

// ---- host: geometry, tables ------------------------------------------------------------------------------------------
constexpr size_t kLds = 160 * 1024;
thread_local int g_mixed_hip = 0;
std::mutex g_mixed_mu;
struct MixedDev { cx<double>* tw2 = nullptr; cx<double>* post = nullptr; };
std::map<std::pair<int, int>, MixedDev> g_mixed;             // (device, N)

bool mixed_geometry(int N, int& r, int& log2p, int max_log2p = 10) {
    if (N < 2 * 3 * 64 || (N & 1)) return false;
    const int M = N / 2;
    for (int rr : {3, 5, 7}) {
        if (M % rr) continue;
        const int P = M / rr;
        if (P & (P - 1)) continue;
        int l = 0; while ((1 << l) < P) ++l;
        if (l < 6 || l > max_log2p) continue;
        r = rr; log2p = l;
        return true;
    }
    return false;
}

int mixed_tables(int N, unit_root_fn unit, MixedTab& mt, int max_log2p = 10) {
    int r = 0, log2p = 0;
    if (!mixed_geometry(N, r, log2p, max_log2p)) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FRAD_E_HIP;
    Tables t;
    const int rc = get_tables(log2p, false, t);
    if (rc != FRAD_OK) return rc;
    std::lock_guard<std::mutex> lk(g_mixed_mu);
    auto key = std::make_pair(dev, N);
    auto it = g_mixed.find(key);
    if (it == g_mixed.end()) {
        const int M = N / 2, P = 1 << log2p;
        std::vector<cx<double>> tw2((size_t)(r - 1) * P), post(2 * ((size_t)M / 2 + 1));
        for (int b = 1; b < r; ++b)
            for (int k1 = 0; k1 < P; ++k1) {
                long double re, im; unit(2LL * b * k1, M, re, im);            // W_M^(b k1) = exp(-i pi 2 b k1 / M)
                tw2[(size_t)(b - 1) * P + k1] = cx<double>{(double)re, (double)im};
            }
        for (int k = 0; k <= M / 2; ++k) {
            long double re, im;
            unit(k, 2LL * N, re, im);                                         // w_k = exp(-i pi k / 2N)
            post[2 * k] = cx<double>{(double)re, (double)im};
            unit((long long)N + 5LL * k, 2LL * N, re, im);                    // g_k = exp(-i pi (1/2 + 5k/2N))
            post[2 * k + 1] = cx<double>{(double)re, (double)im};
        }
        MixedDev d;
        if (hipMalloc(reinterpret_cast<void**>(&d.tw2), tw2.size() * sizeof(cx<double>)) != hipSuccess) return FRAD_E_NOMEM;
        if (hipMalloc(reinterpret_cast<void**>(&d.post), post.size() * sizeof(cx<double>)) != hipSuccess) { (void)hipFree(d.tw2); return FRAD_E_NOMEM; }
        if (hipMemcpy(d.tw2, tw2.data(), tw2.size() * sizeof(cx<double>), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d.post, post.data(), post.size() * sizeof(cx<double>), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d.tw2); (void)hipFree(d.post); return FRAD_E_HIP;
        }
        it = g_mixed.emplace(key, d).first;
    }
    mt.tw2 = it->second.tw2; mt.post = it->second.post; mt.twp = static_cast<const cx<double>*>(t.tw); mt.r = r; mt.log2p = log2p;
    return 1;
}
bool mixed_off() { static const bool d = [] { const char* e = tune("FRAD_TUNE_NO_MIXED"); return e && e[0] == '1'; }(); return d; }

}  // namespace

void mixed_clear() {
    std::lock_guard<std::mutex> lk(g_mixed_mu);
    for (auto& kv : g_mixed) { (void)hipFree(kv.second.tw2); (void)hipFree(kv.second.post); }
    g_mixed.clear();
}
int mixed_last_hip_error() { return g_mixed_hip; }
int mixed_prepare(int N, unit_root_fn unit) { MixedTab mt; const int r = mixed_tables(N, unit, mt); return r < 0 ? r : FRAD_OK; }

#define MCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_mixed_hip = (int)e_; return FRAD_E_HIP; } } while (0)

// 1 = launched, 0 = not this family's geometry (the caller goes on to Bluestein / the direct kernels), < 0 = FRAD_E_*
int launch_p0_fwd_mixed(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax, const Geom& g,
                        int aligned_in, int aligned_out, unit_root_fn unit) {
    if (mixed_off() || g.n_frames > 0x7fffffffLL) return 0;
    const size_t lds = 2 * (size_t)g.N * g.C * 8;
    if (lds > kLds) return 0;
    MixedTab mt;
    const int r = mixed_tables(g.N, unit, mt);
    if (r <= 0) return r;
    if (absmax) MCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)g.n_frames, s));     // atomicMax target
    Geom gg = g; gg.fpb = 1;
    const dim3 grid((unsigned)g.n_frames), blk(256);
#define GO(LGV) do { allow_lds(k_p0_fwd_mixed<LGV>, lds); hipLaunchKernelGGL((k_p0_fwd_mixed<LGV>), grid, blk, lds, s, pcm, pay, absmax, gg, mt, aligned_in, aligned_out); } while (0)
    switch (lg) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; default: GO(3); break; }
#undef GO
    MCHK(hipGetLastError());
    return 1;
}

int launch_p0_inv_mixed(hipStream_t s, const unsigned char* pay, double* out, const Geom& g, int aligned_in, unit_root_fn unit) {
    if (mixed_off() || g.n_frames > 0x7fffffffLL) return 0;
    const size_t lds = 2 * (size_t)g.N * g.C * 8;
    if (lds > kLds) return 0;
    MixedTab mt;
    const int r = mixed_tables(g.N, unit, mt);
    if (r <= 0) return r;
    Geom gg = g; gg.fpb = 1;
    allow_lds(k_p0_inv_mixed<0>, lds);
    hipLaunchKernelGGL(k_p0_inv_mixed<0>, dim3((unsigned)g.n_frames), dim3(256), lds, s, pay, out, gg, mt, aligned_in);
    MCHK(hipGetLastError());
    return 1;
}

int launch_p1_fwd_mixed(int lg, hipStream_t s, const unsigned char* pcm, int32_t* q, int32_t* tq, const Geom& g, const P1Tables& tb,
                        int aligned_in, unit_root_fn unit) {
    if (mixed_off() || g.n_frames > 0x7fffffffLL) return 0;
    int cg = g.C;                                             // channels per pass: as many as the LDS holds twice over + the quantiser's scratch
    while (cg > 0 && 2 * (size_t)g.N * cg * 8 + p1_scratch_bytes(cg, g.N) > kLds) --cg;
    if (cg < 1) return 0;
    const size_t lds = 2 * (size_t)g.N * cg * 8 + p1_scratch_bytes(cg, g.N);
    MixedTab mt;
    const int r = mixed_tables(g.N, unit, mt);
    if (r <= 0) return r;
    Geom gg = g; gg.fpb = 1; gg.cg = cg;
    const dim3 grid((unsigned)g.n_frames), blk(256);
#define GO(LGV) do { allow_lds(k_p1_fwd_mixed<LGV>, lds); hipLaunchKernelGGL((k_p1_fwd_mixed<LGV>), grid, blk, lds, s, pcm, q, tq, gg, tb, mt, aligned_in); } while (0)
    switch (lg) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; default: GO(3); break; }
#undef GO
    MCHK(hipGetLastError());
    return 1;
}

int launch_p1_inv_mixed(hipStream_t s, const int32_t* q, const int32_t* tq, double* out, const Geom& g, const P1Tables& tb, unit_root_fn unit) {
    if (mixed_off() || g.n_frames > 0x7fffffffLL) return 0;
    int cg = g.C;
    while (cg > 0 && 2 * (size_t)g.N * cg * 8 + p1_scratch_bytes(cg, g.N) > kLds) --cg;
    if (cg < 1) return 0;
    const size_t lds = 2 * (size_t)g.N * cg * 8 + p1_scratch_bytes(cg, g.N);
    MixedTab mt;
    const int r = mixed_tables(g.N, unit, mt);
    if (r <= 0) return r;
    Geom gg = g; gg.fpb = 1; gg.cg = cg;
    allow_lds(k_p1_inv_mixed<0>, lds);
    hipLaunchKernelGGL(k_p1_inv_mixed<0>, dim3((unsigned)g.n_frames), dim3(256), lds, s, q, tq, out, gg, tb, mt);
    MCHK(hipGetLastError());
    return 1;
}


// The DCT of `rows` planar float64 rows (forward: x -> X, inverse: X -> x) through the workspace `zw` (rows * N / 2 complex);
// element o of row r goes to out[(r / C) * fstride + (r % C) * cstride + o * ostride] as in k_g_dct (frad_global.hip).
// 1 = done, 0 = N is not of this family, < 0 = FRAD_E_*
int global_dct_mixed(bool fwd, const double* in, double* out, void* zw, int N, int C, long long rows, long long fstride, long long cstride,
                     long long ostride, hipStream_t s, unit_root_fn unit) {
    if (mixed_off() || rows > 65535) return 0;
    MixedTab mt;
    const int rc = mixed_tables(N, unit, mt, 13);
    if (rc <= 0) return rc;
    const int P = 1 << mt.log2p, M = N / 2;
    cx<double>* z = static_cast<cx<double>*>(zw);
    auto sub = [&](auto tag, auto inv) {
        constexpr int L2 = decltype(tag)::value; constexpr bool INV = decltype(inv)::value;
        allow_lds(k_gm_sub<L2, INV>, (size_t)16 << L2);
        hipLaunchKernelGGL((k_gm_sub<L2, INV>), dim3((unsigned)(rows * mt.r)), dim3(Plan<L2>::TEAM), (size_t)16 << L2, s, z, mt.twp);
    };
    auto sub_any = [&](auto inv) {
        switch (mt.log2p) {
            case 6: sub(std::integral_constant<int, 6>{}, inv); break;   case 7: sub(std::integral_constant<int, 7>{}, inv); break;
            case 8: sub(std::integral_constant<int, 8>{}, inv); break;   case 9: sub(std::integral_constant<int, 9>{}, inv); break;
            case 10: sub(std::integral_constant<int, 10>{}, inv); break; case 11: sub(std::integral_constant<int, 11>{}, inv); break;
            case 12: sub(std::integral_constant<int, 12>{}, inv); break; default: sub(std::integral_constant<int, 13>{}, inv); break;
        }
    };
    auto bx = [](int items) { const int b = (items + 255) / 256; return (unsigned)(b < 1 ? 1 : b > 64 ? 64 : b); };
    if (fwd) {
        hipLaunchKernelGGL(k_gm_pack, dim3(bx(M / 2), (unsigned)rows), dim3(256), 0, s, in, z, N, mt.r, mt.log2p);
        sub_any(std::false_type{});
#define GO(RR) hipLaunchKernelGGL(k_gm_radix_post<RR>, dim3(bx(P / 2 + 1), (unsigned)rows), dim3(256), 0, s, z, out, mt, N, C, fstride, cstride, ostride)
        if (mt.r == 3) GO(3); else if (mt.r == 5) GO(5); else GO(7);
#undef GO
    } else {
        hipLaunchKernelGGL(k_gm_pre_inverse, dim3(bx(M / 2 + 1), (unsigned)rows), dim3(256), 0, s, in, z, mt, N);
        sub_any(std::true_type{});
#define GO(RR) hipLaunchKernelGGL(k_gm_radix_unpack<RR>, dim3(bx(P), (unsigned)rows), dim3(256), 0, s, z, out, mt, N, C, fstride, cstride, ostride)
        if (mt.r == 3) GO(3); else if (mt.r == 5) GO(5); else GO(7);
#undef GO
    }
    if (hipGetLastError() != hipSuccess) return FRAD_E_HIP;
    return 1;
}

}  // namespace frad
