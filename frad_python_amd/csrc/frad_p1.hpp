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
#include "frad_wave.hpp"
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
    int f32;                     // float32 / float16 PCM: the reference does not widen it (pcmformat.py:35), so its DCT
                                 // and band energies are float32 and only the divide + quantiser are float64
};

// frad_global.hip: profile 1 through HBM workspaces (frames wider than a CU's LDS at a non-power-of-two size)
int global_p1_analogue(const unsigned char* pcm, int32_t* q, int32_t* tq, const Geom& g, const P1Tables& tb, hipStream_t s);
int global_p1_digital(const int32_t* q, const int32_t* tq, double* out, const Geom& g, const P1Tables& tb, hipStream_t s);

__device__ __forceinline__ double wave_sum_f64(double v) {
    return u2d(wave_allreduce_u64(d2u(v), [](u64 a, u64 b) { return d2u(u2d(a) + u2d(b)); }));
}

// |x|^0.75 with sign (p1tools.py:43) and its inverse |x|^(1/0.75) (p1tools.py:44).  a^0.75 = sqrt(a) * sqrt(sqrt(a)):
// two correctly rounded square roots and one product (<= 1.5 ulp) instead of a generic pow -- the value is rounded
// to an integer right after, so only results within ~1e-16 of a half-integer could differ from pow's.
__device__ __forceinline__ double p1_quant(double x) {
    const double a = fabs(x), r = sqrt(a);
    return copysign(r * sqrt(r), x) * (a != 0.0);
}
// |x|^(4/3) = |x| * cbrt(|x|) (the reference raises to 1/0.75, whose double differs from 4/3 by 2^-54 relative: far
// below the 1e-9 the decode contract states)
__device__ __forceinline__ double p1_dequant(double x) { const double a = fabs(x); return copysign(a * cbrt(a), x) * (a != 0.0); }

// LDS scratch of the quantiser stages, after the transform buffers:
//   thres[cf][27] | step[cf][27] | floor[28] (doubles) | edge[32] (ints, clipped to N) | band_of[N] (bytes)
// The per-launch tables arrive as kernel arguments; p1_tables_to_lds copies them once per block so that the
// data-dependent lookups below are LDS reads (a by-value struct indexed at run time lives in scratch memory).
__host__ __device__ constexpr int p1_scratch_bytes(int cfs, int N) { return cfs * P1_BANDS * 16 + 28 * 8 + 32 * 4 + ((N + 15) / 16) * 16; }

struct P1Lds {
    double* thres; double* step; double* floor_; int* edge; unsigned char* band;
};
__device__ __forceinline__ P1Lds p1_lds(unsigned char* scratch, int cfs) {
    P1Lds l;
    l.thres = reinterpret_cast<double*>(scratch);
    l.step = l.thres + cfs * P1_BANDS;
    l.floor_ = l.step + cfs * P1_BANDS;
    l.edge = reinterpret_cast<int*>(l.floor_ + 28);
    l.band = reinterpret_cast<unsigned char*>(l.edge + 32);
    return l;
}
__device__ __forceinline__ void p1_tables_to_lds(unsigned char* scratch, int cfs, const P1Tables& tb, int N) {
    const P1Lds l = p1_lds(scratch, cfs);
    if (threadIdx.x <= P1_BANDS) l.edge[threadIdx.x] = tb.edge[threadIdx.x] < N ? tb.edge[threadIdx.x] : N;
    if (threadIdx.x < P1_BANDS) l.floor_[threadIdx.x] = tb.floor_[threadIdx.x];
    if ((N & 3) == 0) {
        for (int i = threadIdx.x; i < N / 4; i += blockDim.x)
            reinterpret_cast<uint32_t*>(l.band)[i] = reinterpret_cast<const uint32_t*>(tb.band_of)[i];
    } else {
        for (int i = threadIdx.x; i < N; i += blockDim.x) l.band[i] = tb.band_of[i];
    }
}

// linear ramp between consecutive band starts, endpoint excluded (np.linspace as mapping_from_opus uses it,
// p1tools.py:35-41): y = start_j + i * step_j with step_j = (thres[j+1] - thres[j]) / (bins in band j), computed once
// per band; bins beyond the last start map to 0.
__device__ __forceinline__ void p1_ramp_steps(const P1Lds& l, int cfs) {
    for (int i = threadIdx.x; i < cfs * P1_BANDS; i += blockDim.x) {
        const int j = i % P1_BANDS;
        double st = 0.0;
        if (j < P1_BANDS - 1) {
            const int num = l.edge[j + 1] - l.edge[j];
            if (num > 0) st = (l.thres[i + 1] - l.thres[i]) / (double)num;
        }
        l.step[i] = st;
    }
}
__device__ __forceinline__ double p1_spread(const P1Lds& l, int cf, int k) {
    const int j = l.band[k];
    if (j >= P1_BANDS - 1) return 0.0;
    const double* thres = l.thres + cf * P1_BANDS;
    const int a = l.edge[j];
    const double i = (double)(k - a), st = l.step[cf * P1_BANDS + j];
    double y = i * st;
    if (st == 0.0) y = (i / (double)(l.edge[j + 1] - a)) * (thres[j + 1] - thres[j]);    // numpy's denormal-safe branch
    return y + thres[j];
}

// K7 epilogue: X[k] of `nfl` frames x `cw` channels sit in LDS (xslot<double, SH>); the channels are c0 .. c0+cw-1 of
// the frame's C (c0 = 0, cw = C unless the frame is transformed a channel group at a time); scratch as above for `cfs` slots.
template <int SH>
__device__ FRAD_NOINLINE void p1_quantise(int smem_off, int scratch_off, int slots, double scale, double loss, int nb_used, const Geom& g,
                                          long long f0, int nfl, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                          int c0, int cw, int cfs, int f32 = 0) {
    FRAD_DYN_SMEM(base);
    unsigned char* smem = base + smem_off;
    const int N = g.N, C = g.C, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const P1Lds l = p1_lds(base + scratch_off, cfs);
    // band energies: one (frame, channel, band) task per wave
    for (int task = wave; task < nfl * cw * P1_BANDS; task += nwaves) {
        const int b = task % P1_BANDS, cf = task / P1_BANDS;
        const int a = l.edge[b], e = l.edge[b + 1];
        double acc = 0.0;
        if (f32) for (int k = a + lane; k < e; k += 64) { const float v = (float)xslot<double, SH>(smem, cf, slots, k) * (float)scale; acc += (double)(v * v); }
        else for (int k = a + lane; k < e; k += 64) { const double v = xslot<double, SH>(smem, cf, slots, k) * scale; acc = fma(v, v, acc); }
        acc = wave_sum_f64(acc);
        if (lane == 0) l.thres[cf * P1_BANDS + b] = acc;
    }
    __syncthreads();
    // energies -> thresholds, one entry per thread (p1tools.py:18-33)
    for (int i = threadIdx.x; i < nfl * cw * P1_BANDS; i += blockDim.x) {
        const int b = i % P1_BANDS;
        double t = 0.0;
        if (b < nb_used) t = p1_band_threshold(l.thres[i], l.edge[b + 1] - l.edge[b], l.floor_[b], loss, f32);
        l.thres[i] = t;
    }
    __syncthreads();
    p1_ramp_steps(l, nfl * cw);
    // quantised thresholds, band-major / channel-minor
    for (int i = threadIdx.x; i < nfl * P1_BANDS * cw; i += blockDim.x) {
        const int fl = i / (P1_BANDS * cw), r = i - fl * P1_BANDS * cw, b = r / cw, j = r - b * cw;
        const double t = l.thres[(fl * cw + j) * P1_BANDS + b];
        const double v = log(t > 1.0 ? t : 1.0) / log(2.718281828459045 / 2);
        tq[(f0 + fl) * (long long)(P1_BANDS * C) + b * C + c0 + j] = (int32_t)rint(copysign(pow(fabs(v), 1.0 / 0.75), v));
    }
    __syncthreads();
    // per-bin divide + power-law quantiser, bin-major / channel-minor
    const int NW = N * cw;
    for (int i = threadIdx.x; i < nfl * NW; i += blockDim.x) {
        const int fl = i / NW, r = i - fl * NW, k = r / cw, j = r - k * cw, cf = fl * cw + j;
        double x = xslot<double, SH>(smem, cf, slots, k);
        if (f32) x = (double)(float)x;
        const double div = p1_spread(l, cf, k);
        q[(f0 + fl) * (long long)N * C + (long long)k * C + c0 + j] = p1w_quantise(x, div, scale);   // float32 where it decides, else exact
    }
}

// K8 prologue: q / tq -> X[k] in LDS (channels c0 .. c0+cw-1, as above).
template <int SH>
__device__ FRAD_NOINLINE void p1_dequantise(int smem_off, int scratch_off, int slots, double scale, const Geom& g,
                                            long long f0, int nfl, const int32_t* __restrict__ q, const int32_t* __restrict__ tq,
                                            int c0, int cw, int cfs) {
    FRAD_DYN_SMEM(base);
    unsigned char* smem = base + smem_off;
    const int N = g.N, C = g.C;
    const P1Lds l = p1_lds(base + scratch_off, cfs);
    for (int i = threadIdx.x; i < nfl * P1_BANDS * cw; i += blockDim.x) {
        const int fl = i / (P1_BANDS * cw), r = i - fl * P1_BANDS * cw, b = r / cw, j = r - b * cw;
        const double t = (double)tq[(f0 + fl) * (long long)(P1_BANDS * C) + b * C + c0 + j];
        l.thres[(fl * cw + j) * P1_BANDS + b] = pow(2.718281828459045 / 2, p1_quant(t));
    }
    __syncthreads();                                         // also orders p1_tables_to_lds (kernel start) before its readers
    p1_ramp_steps(l, nfl * cw);
    __syncthreads();
    const int NW = N * cw;
    for (int i = threadIdx.x; i < nfl * NW; i += blockDim.x) {
        const int fl = i / NW, r = i - fl * NW, k = r / cw, j = r - k * cw, cf = fl * cw + j;
        const double v = p1_dequant((double)q[(f0 + fl) * (long long)N * C + (long long)k * C + c0 + j]) / scale;
        xslot<double, SH>(smem, cf, slots, k) = v * p1_spread(l, cf, k);
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
    p1_tables_to_lds(smem + g.fpb * g.C * SLOTS * 16, g.fpb * g.C, tb, g.N);
    stage_in_pcm<double, LG, SH, true>(pcm, 0, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    fft_team<double, LOG2M, false>(buf, t, tw);
    dct_post<double, LOG2M>(buf, t, post);
    __syncthreads();
    p1_quantise<SH>(0, g.fpb * g.C * SLOTS * 16, SLOTS, tb.scale, tb.loss, tb.nb_used, g, f0, nfl, q, tq, 0, g.C, g.fpb * g.C, tb.f32);
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
    p1_tables_to_lds(smem + g.fpb * g.C * SLOTS * 16, g.fpb * g.C, tb, g.N);
    p1_dequantise<SH>(0, g.fpb * g.C * SLOTS * 16, SLOTS, tb.scale, g, f0, nfl, q, tq, 0, g.C, g.fpb * g.C);
    __syncthreads();
    dct_pre_inverse<double, LOG2M>(buf, t, post);
    fft_team<double, LOG2M, true>(buf, t, tw);
    __syncthreads();
    store_pcm_f64<SH, true>(0, out, g, f0, nfl, SLOTS);
}

// The frames the table-driven wave kernel (k_p1_inv_wave) marked: redo[0] = count, redo[1 + i] = frame index.  One frame per
// block pass (launch with g.fpb = 1), the exact arithmetic of k_p1_inv; runs behind the wave kernel on the same stream and
// overwrites its output for those frames.
template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 2 : 1))
k_p1_inv_redo(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
              const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g, P1Tables tb, const int* __restrict__ redo) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const int count = redo[0];
    if ((int)blockIdx.x >= count) return;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    p1_tables_to_lds(smem + g.C * SLOTS * 16, g.C, tb, g.N);
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const long long f0 = redo[1 + i];
        p1_dequantise<SH>(0, g.C * SLOTS * 16, SLOTS, tb.scale, g, f0, 1, q, tq, 0, g.C, g.C);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);
        dct_pre_inverse<double, LOG2M>(buf, tt, post);
        fft_team<double, LOG2M, true>(buf, tt, tw);
        __syncthreads();
        store_pcm_f64<SH, true>(0, out, g, f0, 1, SLOTS);
        __syncthreads();
    }
}

// Frames whose channels exceed a CU's LDS: one frame per block, g.cg channels per pass (stage / transform / quantise
// per group; the quantiser works channel by channel, so groups are independent).
template <int LOG2M, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_p1_fwd_grp(const unsigned char* __restrict__ pcm, int32_t* __restrict__ q, int32_t* __restrict__ tq,
             const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g, P1Tables tb) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM, cg = g.cg;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const int scratch_off = cg * SLOTS * 16;
    p1_tables_to_lds(smem + scratch_off, cg, tb, g.N);
    for (int c0 = 0; c0 < g.C; c0 += cg) {
        const int cgn = g.C - c0 < cg ? g.C - c0 : cg;
        stage_in_pcm_group<double, LG, SH>(pcm, 0, g, f0, SLOTS, c0, cgn);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);
        fft_team<double, LOG2M, false>(buf, tt, tw);
        dct_post<double, LOG2M>(buf, tt, post);
        __syncthreads();
        p1_quantise<SH>(0, scratch_off, SLOTS, tb.scale, tb.loss, tb.nb_used, g, f0, 1, q, tq, c0, cgn, cg, tb.f32);
        __syncthreads();
    }
}

template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_p1_inv_grp(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
             const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g, P1Tables tb) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM, cg = g.cg;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    const int scratch_off = cg * SLOTS * 16;
    p1_tables_to_lds(smem + scratch_off, cg, tb, g.N);
    for (int c0 = 0; c0 < g.C; c0 += cg) {
        const int cgn = g.C - c0 < cg ? g.C - c0 : cg;
        p1_dequantise<SH>(0, scratch_off, SLOTS, tb.scale, g, f0, 1, q, tq, c0, cgn, cg);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);
        dct_pre_inverse<double, LOG2M>(buf, tt, post);
        fft_team<double, LOG2M, true>(buf, tt, tw);
        __syncthreads();
        store_pcm_group<SH>(0, out, g, f0, SLOTS, c0, cgn);
        __syncthreads();
    }
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
    p1_tables_to_lds(smem + 2 * N * C * 8, C, tb, N);
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
    p1_quantise<-1>(N * C * 8, 2 * N * C * 8, N, tb.scale, tb.loss, tb.nb_used, g, f0, 1, q, tq, 0, C, C, tb.f32);
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p1_inv_direct(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ out,
                                                       const double* __restrict__ ct, Geom g, P1Tables tb) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* X = reinterpret_cast<double*>(smem);
    double* x = X + (long long)N * C;
    p1_tables_to_lds(smem + 2 * N * C * 8, C, tb, N);
    p1_dequantise<-1>(0, 2 * N * C * 8, N, tb.scale, g, f0, 1, q, tq, 0, C, C);
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
// element i of the overlap-added output [n_frames, cut, C] (decoder.py:31-38, backend/__init__.py:3)
__device__ __forceinline__ double p1_ola_value(const double* __restrict__ frames, const double* __restrict__ prev_tail, int N, int C, int cut, long long i) {
    const int L = N - cut;
    const double pi = 3.141592653589793;
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
    return v;
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p1_ola(const double* __restrict__ frames, long long n_frames, int N, int C, int cut,
                                                const double* __restrict__ prev_tail, double* __restrict__ out,
                                                double* __restrict__ next_tail) {
    const int L = N - cut;
    const long long total = n_frames * (long long)cut * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        out[i] = p1_ola_value(frames, prev_tail, N, C, cut, i);
    if (next_tail != nullptr && n_frames > 0)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)L * C; i += (long long)gridDim.x * blockDim.x)
            next_tail[i] = frames[((n_frames - 1) * N + cut) * C + i];
}

}  // namespace frad
