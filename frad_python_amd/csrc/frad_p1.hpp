// frad_p1.hpp -- profile 1 (psychoacoustic quantiser) kernels K7 / K8 for gfx950.
//
//   K7  k_p1_fwd / k_p1_fwd_direct   to_f64 + zero-pad + DCT-II + 27-band masking thresholds + per-bin
//                                    divide + power-law quantise -> int32 q[N, C], tq[27, C]
//                                    (fourier/profile1.py:15-40, tools/p1tools.py:15-44)
//   K8  k_p1_inv / k_p1_inv_direct   dequantise + threshold spreading + inverse DCT -> float64 [N, C]
//                                    (fourier/profile1.py:65-77)
//       k_p1_ola                     decoder cross-fade of consecutive frames (decoder.py:28-46)
// Float64 throughout, like the reference on integer / f64 PCM.  Exp-Golomb + deflate stay on the host.
#pragma once
#include "frad_kernels.hpp"
#include <math.h>

namespace frad {

constexpr int P1_BANDS = 27;

// per launch constants, built on the host (see frad_p1.hip)
struct P1Tables {
    const unsigned char* band_of;  // device table [N]: band j with start_j <= k < start_{j+1} (j <= 25), 255 = none
    int edge[P1_BANDS + 1];      // band edges in bins, NOT clipped (p1tools.py:15-16)
    double floor_[P1_BANDS];     // min(ATH(band centre), 1.0)            (p1tools.py:25-31)
    double scale;                // 2^(bits-1)                            (profile1.py:9-10)
    double loss;                 // max(|loss_level|, 0.125)              (profile1.py:20)
    int nb_used;                 // bands before the first empty one      (p1tools.py:22: break)
};

__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += u2d(__shfl_xor(d2u(v), off, 64));
    return v;
}

// |x|^0.75 with sign (p1tools.py:43) and its inverse |x|^(1/0.75) (p1tools.py:44).  a^0.75 = sqrt(a) * sqrt(sqrt(a)):
// two correctly rounded square roots and one product (<= 1.5 ulp) instead of a generic pow -- the value is rounded
// to an integer right after, so only results within ~1e-16 of a half-integer could differ from pow's.
__device__ __forceinline__ double p1_quant(double x) {
    const double a = fabs(x), r = sqrt(a);
    return copysign(r * sqrt(r), x) * (a != 0.0);
}
__device__ __forceinline__ double p1_dequant(double x) { const double a = fabs(x); return copysign(pow(a, 1.0 / 0.75), x) * (a != 0.0); }

// linear ramp between consecutive band starts, endpoint excluded (np.linspace as mapping_from_opus
// uses it, p1tools.py:35-41); bins beyond the last start map to 0.
__device__ __forceinline__ double p1_spread(const double* thres, const int* edge, const unsigned char* band_of, int N, int k) {
    const int j = band_of[k];                             // host-built: start_j <= k < start_{j+1}, starts clipped to N
    if (j >= P1_BANDS - 1) return 0.0;
    const int a = edge[j] < N ? edge[j] : N, e = edge[j + 1] < N ? edge[j + 1] : N;
    const double start = thres[j], delta = thres[j + 1] - start, num = (double)(e - a);
    const double step = delta / num;
    const double i = (double)(k - a);
    const double y = (step == 0.0) ? (i / num) * delta : i * step;       // numpy's denormal-safe branch
    return y + start;
}

// K7 epilogue: X[k] of `nfl` frames x C channels sit in LDS (xslot<double, SH>); scratch = 27*C*nfl doubles.
template <int SH>
__device__ FRAD_NOINLINE void p1_quantise(int smem_off, int scratch_off, int slots, const P1Tables tb, const unsigned char* __restrict__ band_of, const Geom& g,
                                          long long f0, int nfl, int32_t* __restrict__ q, int32_t* __restrict__ tq) {
    FRAD_DYN_SMEM(base);
    unsigned char* smem = base + smem_off;
    double* thres = reinterpret_cast<double*>(base + scratch_off);
    const int N = g.N, C = g.C, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    // band energies: one (frame, channel, band) task per wave
    for (int task = wave; task < nfl * C * P1_BANDS; task += nwaves) {
        const int b = task % P1_BANDS, cf = task / P1_BANDS;
        const int a = tb.edge[b] < N ? tb.edge[b] : N, e = tb.edge[b + 1] < N ? tb.edge[b + 1] : N;
        double acc = 0.0;
        for (int k = a + lane; k < e; k += 64) { const double v = xslot<double, SH>(smem, cf, slots, k) * tb.scale; acc = fma(v, v, acc); }
        acc = wave_sum_f64(acc);
        if (lane == 0) {
            double t = 0.0;
            if (b < tb.nb_used) {
                const double sfq = pow(sqrt(acc / (double)(e - a)), 0.8);
                t = (sfq > tb.floor_[b] ? sfq : tb.floor_[b]) * tb.loss;
            }
            thres[cf * P1_BANDS + b] = t;
        }
    }
    __syncthreads();
    // quantised thresholds, band-major / channel-minor
    for (int i = threadIdx.x; i < nfl * P1_BANDS * C; i += blockDim.x) {
        const int fl = i / (P1_BANDS * C), r = i - fl * P1_BANDS * C, b = r / C, c = r - b * C;
        const double t = thres[(fl * C + c) * P1_BANDS + b];
        const double v = log(t > 1.0 ? t : 1.0) / log(2.718281828459045 / 2);
        tq[(f0 + fl) * (long long)(P1_BANDS * C) + r] = (int32_t)rint(p1_dequant(v));
    }
    // per-bin divide + power-law quantiser, bin-major / channel-minor
    const int NC = N * C;
    for (int i = threadIdx.x; i < nfl * NC; i += blockDim.x) {
        const int fl = i / NC, r = i - fl * NC, k = r / C, c = r - k * C;
        const double x = xslot<double, SH>(smem, fl * C + c, slots, k);
        const double div = p1_spread(thres + (fl * C + c) * P1_BANDS, tb.edge, band_of, N, k);
        const double m = (div == 0.0) ? 0.0 * x : x / div;                // x / inf keeps the sign of x
        q[(f0 + fl) * (long long)NC + r] = (int32_t)rint(p1_quant(m * tb.scale));
    }
}

// K8 prologue: q / tq -> X[k] in LDS.
template <int SH>
__device__ FRAD_NOINLINE void p1_dequantise(int smem_off, int scratch_off, int slots, const P1Tables tb, const unsigned char* __restrict__ band_of, const Geom& g,
                                            long long f0, int nfl, const int32_t* __restrict__ q, const int32_t* __restrict__ tq) {
    FRAD_DYN_SMEM(base);
    unsigned char* smem = base + smem_off;
    double* thres = reinterpret_cast<double*>(base + scratch_off);
    const int N = g.N, C = g.C;
    for (int i = threadIdx.x; i < nfl * P1_BANDS * C; i += blockDim.x) {
        const int fl = i / (P1_BANDS * C), r = i - fl * P1_BANDS * C, b = r / C, c = r - b * C;
        const double t = (double)tq[(f0 + fl) * (long long)(P1_BANDS * C) + r];
        thres[(fl * C + c) * P1_BANDS + b] = pow(2.718281828459045 / 2, p1_quant(t));
    }
    __syncthreads();
    const int NC = N * C;
    for (int i = threadIdx.x; i < nfl * NC; i += blockDim.x) {
        const int fl = i / NC, r = i - fl * NC, k = r / C, c = r - k * C;
        const double v = p1_dequant((double)q[(f0 + fl) * (long long)NC + r]) / tb.scale;
        xslot<double, SH>(smem, fl * C + c, slots, k) = v * p1_spread(thres + (fl * C + c) * P1_BANDS, tb.edge, band_of, N, k);
    }
}

template <int LOG2M, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 2 : 1))
k_p1_fwd(const unsigned char* __restrict__ pcm, int32_t* __restrict__ q, int32_t* __restrict__ tq,
         const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g, P1Tables tb, int aligned_in) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    stage_in_pcm<double, LG, SH, true>(pcm, 0, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    fft_team<double, LOG2M, false>(buf, t, tw);
    dct_post<double, LOG2M>(buf, t, post);
    __syncthreads();
    p1_quantise<SH>(0, g.fpb * g.C * SLOTS * 16, SLOTS, tb, tb.band_of, g, f0, nfl, q, tq);
}

template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 2 : 1))
k_p1_inv(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
         const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g, P1Tables tb) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    p1_dequantise<SH>(0, g.fpb * g.C * SLOTS * 16, SLOTS, tb, tb.band_of, g, f0, nfl, q, tq);
    __syncthreads();
    dct_pre_inverse<double, LOG2M>(buf, t, post);
    fft_team<double, LOG2M, true>(buf, t, tw);
    __syncthreads();
    store_pcm_f64<SH, true>(0, out, g, f0, nfl, SLOTS);
}

// any legal compact frame size (160/192/224 * 2^n ...): direct cosine sums, one frame per block
template <int LG>
__global__ void __launch_bounds__(256) k_p1_fwd_direct(const unsigned char* __restrict__ pcm, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                                       const double* __restrict__ ct, Geom g, P1Tables tb, int aligned_in) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* x = reinterpret_cast<double*>(smem);
    double* X = x + (long long)N * C;
    stage_in_pcm<double, LG, -1, false>(pcm, 0, g, f0, 1, N, aligned_in != 0);
    __syncthreads();
    const double inv_n = 1.0 / (double)N;
    const unsigned fourN = 4u * (unsigned)N;
    for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
        const int c = i / N, k = i - c * N;
        const double* xc = x + (long long)c * N;
        double acc = 0.0;
        unsigned j = (unsigned)k % fourN;
        const unsigned step = (2u * (unsigned)k) % fourN;
        for (int n = 0; n < N; ++n) { acc = fma(xc[n], ct[j], acc); j += step; if (j >= fourN) j -= fourN; }
        X[(long long)c * N + k] = acc * inv_n;
    }
    __syncthreads();
    p1_quantise<-1>(N * C * 8, 2 * N * C * 8, N, tb, tb.band_of, g, f0, 1, q, tq);
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p1_inv_direct(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
                                                       const double* __restrict__ ct, Geom g, P1Tables tb) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* X = reinterpret_cast<double*>(smem);
    double* x = X + (long long)N * C;
    p1_dequantise<-1>(0, 2 * N * C * 8, N, tb, tb.band_of, g, f0, 1, q, tq);
    __syncthreads();
    const unsigned fourN = 4u * (unsigned)N;
    for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
        const int c = i / N, n = i - c * N;
        const double* Xc = X + (long long)c * N;
        double acc = 0.0;
        const unsigned step = (2u * (unsigned)n + 1u) % fourN;
        unsigned j = step;
        for (int k = 1; k < N; ++k) { acc = fma(Xc[k], ct[j], acc); j += step; if (j >= fourN) j -= fourN; }
        x[(long long)c * N + n] = Xc[0] + 2.0 * acc;
    }
    __syncthreads();
    store_pcm_f64<-1, false>(N * C * 8, out, g, f0, 1, N);
}

// R8: decoder overlap-add over a batch of consecutive frames (decoder.py:28-46).  Frame i keeps
// rows [0, cut) -- its first L = N - cut rows cross-faded with frame i-1's tail -- and hands its
// own tail [cut, N) to frame i+1.  out: [n_frames, cut, C]; next_tail: [L, C].
template <int UNUSED>
__global__ void __launch_bounds__(256) k_p1_ola(const double* __restrict__ frames, long long n_frames, int N, int C, int cut,
                                                const double* __restrict__ prev_tail, double* __restrict__ out,
                                                double* __restrict__ next_tail) {
    const int L = N - cut;
    const long long total = n_frames * (long long)cut * C;
    const double pi = 3.141592653589793;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / ((long long)cut * C);
        const int r = (int)(i - f * (long long)cut * C), n = r / C, c = r - n * C;
        double v = frames[(f * N + n) * C + c];
        if (n < L) {
            const double* tail = f > 0 ? frames + ((f - 1) * N + cut) * C : prev_tail;
            if (tail != nullptr) {
                // hanning_in_overlap (backend/__init__.py:3): w[i] = 0.5 (1 - cos(pi (i+1) / (L+1)))
                const double w_in = 0.5 * (1.0 - cos(pi * (double)(n + 1) / (double)(L + 1)));
                const double w_out = 0.5 * (1.0 - cos(pi * (double)(L - n) / (double)(L + 1)));
                v = v * w_in;
                v = v + tail[(long long)n * C + c] * w_out;
            }
        }
        out[i] = v;
    }
    if (next_tail != nullptr && n_frames > 0)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)L * C; i += (long long)gridDim.x * blockDim.x)
            next_tail[i] = frames[((n_frames - 1) * N + cut) * C + i];
}

}  // namespace frad
