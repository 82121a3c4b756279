// frad_global.hip -- the path of last resort: frames that no LDS-resident kernel can hold.
//
// Every transform kernel of this library keeps a frame's channels in the 160 KiB LDS of one CU.  A frame can be larger
// than that: the clip's last frame is whatever is left (encoder.py:72-93), lossless frame sizes go up to 2^32 - 1, and
// the compact table reaches 28 672 samples (fourier/profiles.py:14-23) -- one float64 channel of that is 224 KiB.  Such
// frames used to be refused with FRAD_E_UNSUPPORTED.  Here they run through HBM workspaces instead:
//
//   encode  PCM -> planar float64 workspace (to_f64, zero padding)            k_g_pcm_to_ws
//           DCT-II of every channel row as a tiled cosine product             k_g_dct<true>     (O(N^2), exact table)
//           profile 0: cast + pack + |X| max = the profile-4 pack kernel on the coefficient rows (frad_p4_analogue)
//           profile 1: band energies, thresholds, quantiser on the rows       k_g_p1_quant
//   decode  profile 0: unpack + scrub = frad_p4_digital into the workspace; profile 1: k_g_p1_dequant
//           inverse DCT                                                       k_g_dct<false>
//
// The arithmetic is the direct kernels' (frad_kernels.hpp direct_product: same table, same summation order per output),
// so the tolerance contract is unchanged.  Throughput is that of a dense product served from L2 -- tens of
// microseconds per channel-frame at N = 28 672 -- which is what an oversized tail frame or the rare 0.6 s compact
// frame costs; nothing is refused any more.  Workspaces come from the stream-ordered allocator and the batch is cut
// into chunks of at most 128 MiB of workspace.
#include "frad_p1.hpp"
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

namespace frad {
namespace {

constexpr int TN = 2048;                               // samples (or bins) of a row staged in LDS per step

// PCM -> workspace.  xw[(f * C + c) * N + n] = to_f64(pcm[f * stride + n][c]) for n < n_valid, 0 beyond.
__global__ void __launch_bounds__(256) k_g_pcm_to_ws(const unsigned char* __restrict__ pcm, double* __restrict__ xw, Geom g, long long f0) {
    const int N = g.N, C = g.C, lg = (g.dtype >> 1) & 3, bx = g.fpb;                 // g.fpb: blocks per frame here
    const long long f = blockIdx.x / bx;
    const unsigned char* src = pcm + (((f0 + f) * g.frame_stride * C) << lg);
    for (long long i = (long long)(blockIdx.x - f * bx) * blockDim.x + threadIdx.x; i < (long long)N * C; i += (long long)bx * blockDim.x) {
        const int n = (int)(i / C), c = (int)(i - (long long)n * C);
        double v = 0.0;
        if (n < g.n_valid) {
            const u64 raw = load_raw(src + (i << lg), lg);
            v = dtype_is_f32_class(g.dtype) ? (double)cvt_pcm<float>(raw, g.dtype, g.raw_be != 0) : cvt_pcm<double>(raw, g.dtype, g.raw_be != 0);
        }
        xw[(f * C + c) * (long long)N + n] = v;
    }
}

// 256 outputs of one row (channel-frame) per block, `bpr` blocks per row.  in: planar rows of N doubles.  out: element o of row
// r goes to out[r_base + o * ostride] with r_base = (r / C) * fstride + (r % C) * cstride (planar or interleaved).
//   FWD: X[k] = (1/N) sum_n x[n] cos(pi k (2n+1) / 2N)          ct[j] = cos(pi j / 2N), j = k (2n + 1) mod 4N
//   INV: x[n] = X[0] + 2 sum_{k>=1} X[k] cos(pi k (2n+1) / 2N)
template <bool FWD>
__global__ void __launch_bounds__(256) k_g_dct(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ ct,
                                               int N, int C, long long fstride, long long cstride, long long ostride, int bpr) {
    FRAD_DYN_SMEM(smem);
    double* tile = reinterpret_cast<double*>(smem);
    const long long r = blockIdx.x / bpr;
    const double* row = in + r * (long long)N;
    const int o = (int)(blockIdx.x - r * bpr) * blockDim.x + threadIdx.x;
    const unsigned long long fourN = 4ull * (unsigned long long)N;
    const unsigned long long step = FWD ? (2ull * (unsigned long long)o) % fourN : (2ull * (unsigned long long)o + 1ull) % fourN;
    double acc = 0.0;
    for (int t0 = 0; t0 < N; t0 += TN) {
        const int tn = N - t0 < TN ? N - t0 : TN;
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += blockDim.x) tile[i] = row[t0 + i];
        __syncthreads();
        if (o < N) {
            // table index of the tile's first term: FWD k (2 t0 + 1), INV t0 (2n + 1), both mod 4N
            unsigned long long j = FWD ? ((unsigned long long)o % fourN + step * (unsigned long long)t0) % fourN
                                       : (step * (unsigned long long)t0) % fourN;
            unsigned jj = (unsigned)j;
            const unsigned st = (unsigned)step, fn = (unsigned)fourN;
            int i = 0;
            if (!FWD && t0 == 0) { i = 1; jj = st; }                         // the k = 0 term is added apart, undoubled
            for (; i < tn; ++i) { acc = fma(tile[i], ct[jj], acc); jj += st; if (jj >= fn) jj -= fn; }
        }
    }
    if (o < N) {
        const double v = FWD ? acc * (1.0 / (double)N) : row[0] + 2.0 * acc;
        out[(r / C) * fstride + (r % C) * cstride + (long long)o * ostride] = v;
    }
}

// profile 1, encode side: one (frame, channel) row per block; X planar in the workspace.  Same formulas as p1_quantise
// (frad_p1.hpp), with the coefficients read from HBM instead of LDS.
__global__ void __launch_bounds__(256) k_g_p1_quant(const double* __restrict__ xw, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                                    Geom g, P1Tables tb, long long f0) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const long long r = blockIdx.x, f = r / C;
    const int c = (int)(r - f * C);
    const double* X = xw + r * (long long)N;
    p1_tables_to_lds(smem, 1, tb, N);
    const P1Lds l = p1_lds(smem, 1);
    __syncthreads();
    for (int b = wave; b < P1_BANDS; b += nwaves) {
        const int a = l.edge[b], e = l.edge[b + 1];
        double acc = 0.0;
        if (tb.f32) for (int k = a + lane; k < e; k += 64) { const float v = (float)X[k] * (float)tb.scale; acc += (double)(v * v); }
        else for (int k = a + lane; k < e; k += 64) { const double v = X[k] * tb.scale; acc = fma(v, v, acc); }
        acc = wave_sum_f64(acc);
        if (lane == 0) l.thres[b] = acc;
    }
    __syncthreads();
    if (threadIdx.x < P1_BANDS) {
        const int b = threadIdx.x;
        double t = 0.0;
        if (b < tb.nb_used) t = p1_band_threshold(l.thres[b], l.edge[b + 1] - l.edge[b], l.floor_[b], tb.loss, tb.f32);
        l.thres[b] = t;
    }
    __syncthreads();
    p1_ramp_steps(l, 1);
    if (threadIdx.x < P1_BANDS) {
        const double t = l.thres[threadIdx.x];
        const double v = log(t > 1.0 ? t : 1.0) / log(2.718281828459045 / 2);
        tq[(f0 + f) * (long long)(P1_BANDS * C) + threadIdx.x * C + c] = (int32_t)rint(copysign(pow(fabs(v), 1.0 / 0.75), v));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const double x = tb.f32 ? (double)(float)X[k] : X[k];
        const double div = p1_spread(l, 0, k);
        q[(f0 + f) * (long long)N * C + (long long)k * C + c] = p1w_quantise(x, div, tb.scale);
    }
}

// profile 1, decode side: q / tq -> X rows in the workspace (p1_dequantise with the rows in HBM)
__global__ void __launch_bounds__(256) k_g_p1_dequant(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, double* __restrict__ xw,
                                                      Geom g, P1Tables tb, long long f0) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long r = blockIdx.x, f = r / C;
    const int c = (int)(r - f * C);
    double* X = xw + r * (long long)N;
    p1_tables_to_lds(smem, 1, tb, N);
    const P1Lds l = p1_lds(smem, 1);
    if (threadIdx.x < P1_BANDS) {
        const double t = (double)tq[(f0 + f) * (long long)(P1_BANDS * C) + threadIdx.x * C + c];
        l.thres[threadIdx.x] = pow(2.718281828459045 / 2, p1_quant(t));
    }
    __syncthreads();
    p1_ramp_steps(l, 1);
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const double v = p1_dequant((double)q[(f0 + f) * (long long)N * C + (long long)k * C + c]) / tb.scale;
        X[k] = v * p1_spread(l, 0, k);
    }
}

thread_local int g_glob_hip = 0;
#define GCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_glob_hip = (int)e_; return FRAD_E_HIP; } } while (0)

struct Ws {                                            // stream-ordered scratch, released on every path out
    hipStream_t s; void* p = nullptr;
    explicit Ws(hipStream_t st) : s(st) {}
    int get(size_t bytes) { return hipMallocAsync(&p, bytes, s) == hipSuccess ? FRAD_OK : FRAD_E_NOMEM; }
    ~Ws() { if (p) (void)hipFreeAsync(p, s); }
};
long long frames_per_chunk(long long n_frames, int N, int C) {
    const long long per = (long long)N * C * 8;
    long long k = (128LL << 20) / per;
    if (k < 1) k = 1;
    return k < n_frames ? k : n_frames;
}
int dct_bpr(int N) { return (N + 255) / 256; }
dim3 dct_grid(int N, long long rows) { return dim3((unsigned)(dct_bpr(N) * rows)); }
int stage_bpf(int N, int C) { const long long b = ((long long)N * C + 255) / 256; return (int)(b > 64 ? 64 : b); }

// DCT of `rows` planar rows: O(N log N) through the mixed-radix kernels (frad_mixed.hip) when N = 2 r 2^p, else the dense product
template <bool FWD>
int rows_dct(const double* in, double* out, void* zw, const double* ct, int N, int C, long long rows, long long fstride, long long cstride,
             long long ostride, hipStream_t s) {
    if (zw != nullptr) {
        const int r = global_dct_mixed(FWD, in, out, zw, N, C, rows, fstride, cstride, ostride, s, unit_root);
        if (r != 0) return r < 0 ? r : FRAD_OK;
    }
    hipLaunchKernelGGL(k_g_dct<FWD>, dct_grid(N, rows), dim3(256), TN * 8, s, in, out, ct, N, C, fstride, cstride, ostride, dct_bpr(N));
    return FRAD_OK;
}
bool mixed_length(int N) {                                   // N = 2 r 2^p, r in {3, 5, 7}, 6 <= p <= 13
    if (N & 1) return false;
    const int M = N / 2;
    for (int r : {3, 5, 7}) if (M % r == 0) { const int P = M / r; if (P >= 64 && P <= 8192 && (P & (P - 1)) == 0) return true; }
    return false;
}

}  // namespace

int global_last_hip_error() { return g_glob_hip; }

// profile 0 encode of frames [0, n_frames) through the workspaces.  `g` as frad_p0_analogue builds it.
int global_p0_analogue(const unsigned char* pcm, unsigned char* payload, double* absmax, const Geom& g, uint32_t flags, hipStream_t s) {
    const int N = g.N, C = g.C;
    if ((long long)N * C * 8 > (1LL << 31)) return FRAD_E_UNSUPPORTED;
    DirectTable d; int rc = get_direct(N, d);
    if (rc != FRAD_OK) return rc;
    const long long chunk = frames_per_chunk(g.n_frames, N, C);
    Ws xw(s), Xw(s), zw(s);
    if ((rc = xw.get((size_t)chunk * N * C * 8)) != FRAD_OK || (rc = Xw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    if (mixed_length(N) && (rc = zw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    for (long long f0 = 0; f0 < g.n_frames; f0 += chunk) {
        const long long nf = g.n_frames - f0 < chunk ? g.n_frames - f0 : chunk;
        Geom gs = g; gs.fpb = stage_bpf(N, C);
        hipLaunchKernelGGL(k_g_pcm_to_ws, dim3((unsigned)(gs.fpb * nf)), dim3(256), 0, s, pcm, static_cast<double*>(xw.p), gs, f0);
        // coefficient rows interleaved [frame][k][c]: the order profile 0 packs them in (freqs.T.ravel(), profile0.py:29)
        rc = rows_dct<true>(static_cast<const double*>(xw.p), static_cast<double*>(Xw.p), zw.p, d.ct, N, C, nf * C, (long long)N * C, 1LL, (long long)C, s);
        if (rc != FRAD_OK) return rc;
        GCHK(hipGetLastError());
        rc = frad_p4_analogue(Xw.p, FRAD_PCM_F64LE, nf, N, C, N, g.bits, flags & FRAD_LITTLE_ENDIAN, payload + f0 * g.payload_stride,
                              g.payload_stride, absmax ? absmax + f0 : nullptr, s);
        if (rc != FRAD_OK) return rc;
    }
    return FRAD_OK;
}

int global_p0_digital(const unsigned char* payload, double* out, const Geom& g, uint32_t flags, hipStream_t s) {
    const int N = g.N, C = g.C;
    if ((long long)N * C * 8 > (1LL << 31)) return FRAD_E_UNSUPPORTED;
    DirectTable d; int rc = get_direct(N, d);
    if (rc != FRAD_OK) return rc;
    const long long chunk = frames_per_chunk(g.n_frames, N, C);
    Ws Xi(s), Xp(s), zw(s);
    if ((rc = Xi.get((size_t)chunk * N * C * 8)) != FRAD_OK || (rc = Xp.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    if (mixed_length(N) && (rc = zw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    for (long long f0 = 0; f0 < g.n_frames; f0 += chunk) {
        const long long nf = g.n_frames - f0 < chunk ? g.n_frames - f0 : chunk;
        rc = frad_p4_digital(payload + f0 * g.payload_stride, g.payload_stride, nf, N, C, g.bits, flags & FRAD_LITTLE_ENDIAN,
                             static_cast<double*>(Xi.p), s);               // unpack + NaN/Inf scrub, [frame][k][c]
        if (rc != FRAD_OK) return rc;
        // de-interleave into rows with the forward kernel's twin: a "transform" of length 1 would do, but the planar
        // copy is simply the inverse kernel's input layout -> read interleaved through a strided gather kernel
        Geom gi = g; gi.dtype = FRAD_PCM_F64LE; gi.raw_be = 0; gi.frame_stride = N; gi.n_valid = N; gi.fpb = stage_bpf(N, C);
        hipLaunchKernelGGL(k_g_pcm_to_ws, dim3((unsigned)(gi.fpb * nf)), dim3(256), 0, s, static_cast<const unsigned char*>(Xi.p),
                           static_cast<double*>(Xp.p), gi, 0LL);
        rc = rows_dct<false>(static_cast<const double*>(Xp.p), out + f0 * (long long)N * C, zw.p, d.ct, N, C, nf * C, (long long)N * C, 1LL, (long long)C, s);
        if (rc != FRAD_OK) return rc;
        GCHK(hipGetLastError());
    }
    return FRAD_OK;
}

int global_p1_analogue(const unsigned char* pcm, int32_t* q, int32_t* tq, const Geom& g, const P1Tables& tb, hipStream_t s) {
    const int N = g.N, C = g.C;
    DirectTable d; int rc = get_direct(N, d);
    if (rc != FRAD_OK) return rc;
    const long long chunk = frames_per_chunk(g.n_frames, N, C);
    Ws xw(s), Xw(s), zw(s);
    if ((rc = xw.get((size_t)chunk * N * C * 8)) != FRAD_OK || (rc = Xw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    if (mixed_length(N) && (rc = zw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    const size_t lds = p1_scratch_bytes(1, N);
    for (long long f0 = 0; f0 < g.n_frames; f0 += chunk) {
        const long long nf = g.n_frames - f0 < chunk ? g.n_frames - f0 : chunk;
        Geom gs = g; gs.fpb = stage_bpf(N, C);
        hipLaunchKernelGGL(k_g_pcm_to_ws, dim3((unsigned)(gs.fpb * nf)), dim3(256), 0, s, pcm, static_cast<double*>(xw.p), gs, f0);
        rc = rows_dct<true>(static_cast<const double*>(xw.p), static_cast<double*>(Xw.p), zw.p, d.ct, N, C, nf * C, (long long)N * C, (long long)N, 1LL, s);   // planar rows for the band sums
        if (rc != FRAD_OK) return rc;
        hipLaunchKernelGGL(k_g_p1_quant, dim3((unsigned)(nf * C)), dim3(256), lds, s, static_cast<const double*>(Xw.p), q, tq, g, tb, f0);
        GCHK(hipGetLastError());
    }
    return FRAD_OK;
}

int global_p1_digital(const int32_t* q, const int32_t* tq, double* out, const Geom& g, const P1Tables& tb, hipStream_t s) {
    const int N = g.N, C = g.C;
    DirectTable d; int rc = get_direct(N, d);
    if (rc != FRAD_OK) return rc;
    const long long chunk = frames_per_chunk(g.n_frames, N, C);
    Ws Xw(s), zw(s);
    if ((rc = Xw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    if (mixed_length(N) && (rc = zw.get((size_t)chunk * N * C * 8)) != FRAD_OK) return rc;
    const size_t lds = p1_scratch_bytes(1, N);
    for (long long f0 = 0; f0 < g.n_frames; f0 += chunk) {
        const long long nf = g.n_frames - f0 < chunk ? g.n_frames - f0 : chunk;
        hipLaunchKernelGGL(k_g_p1_dequant, dim3((unsigned)(nf * C)), dim3(256), lds, s, q, tq, static_cast<double*>(Xw.p), g, tb, f0);
        rc = rows_dct<false>(static_cast<const double*>(Xw.p), out + f0 * (long long)N * C, zw.p, d.ct, N, C, nf * C, (long long)N * C, 1LL, (long long)C, s);
        if (rc != FRAD_OK) return rc;
        GCHK(hipGetLastError());
    }
    return FRAD_OK;
}

}  // namespace frad
