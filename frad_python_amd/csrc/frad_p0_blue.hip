// frad_p0_blue.hip -- profile 0 for frame lengths that are not a power of two (a clip's last, short
// frame: encoder.py:72-93 cuts whatever is left), in O(N log N).
//
// DCT-II of any length N = Makhoul's permutation v (v[i] = x[2i], v[N-1-i] = x[2i+1]), an N-point DFT V of
// v, and X[k] = Re(e^{-i pi k / 2N} V[k]) / N.  The DFT is Bluestein's: with w_n = e^{i pi n^2 / N},
//   V[k] = conj(w_k) * sum_n (v[n] conj(w_n)) w_{k-n},
// a linear convolution, done cyclically at L = 2^p >= 2N - 1 with the LDS-resident power-of-two FFT the
// other kernels use (fft_team): FFT_L(a) * FFT_L(b) / L -> inverse FFT.  The inverse transform is the same
// DFT applied to conj(A), A[k] = e^{i pi k / 2N} (X[k] - i X[N-k]), whose real part is v again.
// Tables (host, long double, exact index reduction): conj(w_n), FFT_L(b) / L, conj(w_k) e^{-i pi k / 2N}.
// float64 only; f32/f16 PCM at such lengths keeps the direct kernels.
#include "../../include/frad_hip.h"
#include "frad_launch.hpp"

#include <cmath>
#include <cstdlib>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace frad {

template <int LOG2L>
__device__ __forceinline__ void blue_convolve(cx<double>* buf, int t, const cx<double>* __restrict__ tw,
                                              const cx<double>* __restrict__ bhat) {
    constexpr int L = 1 << LOG2L, TEAM = Plan<LOG2L>::TEAM, SH = Plan<LOG2L>::SH;
    fft_team<double, LOG2L, false>(buf, t, tw);
#pragma unroll
    for (int i = 0; i < L / TEAM; ++i) {
        const int slot = t + i * TEAM;
        cx<double>& z = buf[phys<double, SH>(slot)];
        z = cmul(z, bhat[slot]);
    }
    team_sync<TEAM>();
    fft_team<double, LOG2L, true>(buf, t, tw);
}

// LDS: g.cg complex buffers of L slots, then the real X / x area [C][N]
template <int LOG2L, int LG>
__global__ void __launch_bounds__(1024) k_p0_fwd_blue(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                      double* absmax, const cx<double>* __restrict__ tw,
                                                      const cx<double>* __restrict__ wconj, const cx<double>* __restrict__ bhat,
                                                      const cx<double>* __restrict__ pw, const cx<double>* __restrict__ pw2, Geom g, int aligned_out) {
    constexpr int L = 1 << LOG2L, TEAM = Plan<LOG2L>::TEAM, SH = Plan<LOG2L>::SH;
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C, cg = g.cg;
    const long long f = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* bufs = reinterpret_cast<cx<double>*>(smem);
    cx<double>* buf = bufs + (long long)cf * L;
    const int xoff = cg * L * 16;
    double* X = reinterpret_cast<double*>(smem + xoff);
    const unsigned char* src = pcm + ((frame_base(g, f) * C) << LG);
    const double inv_n = 1.0 / (double)N;
    const bool whole = g.in_mode != 0;                       // X area holds all C channels: one whole-frame pack at the end
    if (g.cc_fast) {
        // Channel PAIRS (whole frames, even C): the two real sequences of channels 2p, 2p+1 ride one complex Bluestein
        // convolution as v0 + i v1 -- half the FFT work and half the LDS -- and are told apart afterwards by the Hermitian
        // symmetry of a real sequence's DFT: with u_k = conj(w_k) z_k,  V0 = (u_k + conj(u_{N-k})) / 2,
        // V1 = (u_k - conj(u_{N-k})) / 2i;  pw2[k] = e^{-i pi k / 2N} w_{N-k} (pw2[0] = 1) is the second half's table.
        const int P = C / 2;
        const double inv_2n = 0.5 * inv_n;
        for (int p0 = 0; p0 < P; p0 += cg) {
            const int cgn = P - p0 < cg ? P - p0 : cg;
            for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {
                const int n = q / cgn, j = q - n * cgn, c = 2 * (p0 + j);
                double v0 = 0.0, v1 = 0.0;
                if (n < g.n_valid) {
                    const unsigned char* e = src + (((long long)n * C + c) << LG);
                    v0 = cvt_pcm<double>(load_raw(e, LG), g.dtype, g.raw_be);
                    v1 = cvt_pcm<double>(load_raw(e + (1 << LG), LG), g.dtype, g.raw_be);
                }
                const int slot = makhoul(n, N);
                const cx<double> w = wconj[slot];
                bufs[(long long)j * L + phys<double, SH>(slot)] = cx<double>{v0 * w.x - v1 * w.y, v0 * w.y + v1 * w.x};
            }
            for (int q = threadIdx.x; q < (L - N) * cg; q += blockDim.x) {
                const int sl = q / cg, j = q - sl * cg;
                bufs[(long long)j * L + phys<double, SH>(N + sl)] = cx<double>{0.0, 0.0};
            }
            __syncthreads();
            blue_convolve<LOG2L>(buf, t, tw, bhat);
            if (cf < cgn) {
                double* X0 = X + (long long)(2 * (p0 + cf)) * N;
                for (int k = t; k < N; k += TEAM) {
                    const cx<double> z = buf[phys<double, SH>(k)], z2 = buf[phys<double, SH>(k ? N - k : 0)];
                    const cx<double> A = cmul(z, pw[k]), B = cmul(cx<double>{z2.x, -z2.y}, pw2[k]);
                    X0[k] = (A.x + B.x) * inv_2n;
                    X0[N + k] = (A.y - B.y) * inv_2n;
                }
            }
            __syncthreads();
        }
        pack_out_any<double, -1>(xoff, payload, absmax, g, f, 1, N, aligned_out != 0);
        return;
    }
    for (int c0 = 0; c0 < C; c0 += cg) {
        const int cgn = C - c0 < cg ? C - c0 : cg;
        {                                                                    // a[slot] = v[slot] conj(w_slot)
            const int total = N * cgn, TH = blockDim.x;
            auto place = [&](int q, u64 raw) {
                const int n = q / cgn, j = q - n * cgn;
                const double v = n < g.n_valid ? cvt_pcm<double>(raw, g.dtype, g.raw_be) : 0.0;
                const int slot = makhoul(n, N);
                const cx<double> w = wconj[slot];
                bufs[(long long)j * L + phys<double, SH>(slot)] = cx<double>{v * w.x, v * w.y};
            };
            auto fetch = [&](int q) -> u64 {
                const int n = q / cgn, j = q - n * cgn;
                return n < g.n_valid ? load_raw(src + (((long long)n * C + c0 + j) << LG), LG) : 0;
            };
            int q = threadIdx.x;
            for (; q + 7 * TH < total; q += 8 * TH) {                         // eight element loads in flight per lane
                u64 raw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) raw[i] = fetch(q + i * TH);
#pragma unroll
                for (int i = 0; i < 8; ++i) place(q + i * TH, raw[i]);
            }
            for (; q < total; q += TH) place(q, fetch(q));
        }
        for (int q = threadIdx.x; q < (L - N) * cg; q += blockDim.x) {        // zero padding up to L
            const int s = q / cg, j = q - s * cg;
            bufs[(long long)j * L + phys<double, SH>(N + s)] = cx<double>{0.0, 0.0};
        }
        __syncthreads();
        blue_convolve<LOG2L>(buf, t, tw, bhat);
        if (cf < cgn) {
            for (int k = t; k < N; k += TEAM) {
                const cx<double> z = buf[phys<double, SH>(k)], p = pw[k];
                X[(long long)((whole ? c0 : 0) + cf) * N + k] = (z.x * p.x - z.y * p.y) * inv_n;
            }
        }
        __syncthreads();
        if (!whole) {                                        // big frames: pack this channel group now, per value
            pack_out_group<double, -1>(xoff, payload, absmax, g, f, N, c0, cgn);
            __syncthreads();
        }
    }
    if (whole) pack_out_any<double, -1>(xoff, payload, absmax, g, f, 1, N, aligned_out != 0);
}

template <int LOG2L>
__global__ void __launch_bounds__(1024) k_p0_inv_blue(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                      const cx<double>* __restrict__ tw, const cx<double>* __restrict__ wconj,
                                                      const cx<double>* __restrict__ bhat, const cx<double>* __restrict__ pw,
                                                      Geom g, int aligned_in) {
    constexpr int L = 1 << LOG2L, TEAM = Plan<LOG2L>::TEAM, SH = Plan<LOG2L>::SH;
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C, cg = g.cg;
    const long long f = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* bufs = reinterpret_cast<cx<double>*>(smem);
    cx<double>* buf = bufs + (long long)cf * L;
    const int xoff = cg * L * 16;
    double* X = reinterpret_cast<double*>(smem + xoff);
    const bool whole = g.in_mode != 0;
    if (whole) unpack_in_any<-1>(payload, xoff, g, f, 1, N, aligned_in != 0);
    __syncthreads();
    const int half = (N + 1) / 2;                            // slots [0, half) hold the even time samples
    if (g.cc_fast) {
        // channel pairs: DFT(conj(A0) + i conj(A1)) = N (v0 + i v1) -- each A is Hermitian, so each transform is real
        const int P = C / 2;
        for (int p0 = 0; p0 < P; p0 += cg) {
            const int cgn = P - p0 < cg ? P - p0 : cg;
            for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {
                const int k = q / cgn, j = q - k * cgn;
                const double* X0 = X + (long long)(2 * (p0 + j)) * N;
                const double* X1 = X0 + N;
                const cx<double> a = {X0[k] - (k > 0 ? X1[N - k] : 0.0), (k > 0 ? X0[N - k] : 0.0) + X1[k]};
                bufs[(long long)j * L + phys<double, SH>(k)] = cmul(a, pw[k]);
            }
            for (int q = threadIdx.x; q < (L - N) * cg; q += blockDim.x) {
                const int sl = q / cg, j = q - sl * cg;
                bufs[(long long)j * L + phys<double, SH>(N + sl)] = cx<double>{0.0, 0.0};
            }
            __syncthreads();                                  // (every X of this pair group has been read)
            blue_convolve<LOG2L>(buf, t, tw, bhat);
            if (cf < cgn) {
                double* x0 = X + (long long)(2 * (p0 + cf)) * N;
                for (int n = t; n < N; n += TEAM) {
                    const cx<double> z = buf[phys<double, SH>(n)], w = wconj[n];
                    const int time = n < half ? 2 * n : 2 * (N - 1 - n) + 1;
                    x0[time] = z.x * w.x - z.y * w.y;
                    x0[N + time] = z.x * w.y + z.y * w.x;
                }
            }
            __syncthreads();
        }
        store_pcm_f64<-1, false>(xoff, out, g, f, 1, N);
        return;
    }
    for (int c0 = 0; c0 < C; c0 += cg) {
        const int cgn = C - c0 < cg ? C - c0 : cg;
        if (!whole) {
            unpack_in_group<-1>(payload, xoff, g, f, N, c0, cgn);
            __syncthreads();
        }
        for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {           // conj(A[k]) conj(w_k)
            const int k = q / cgn, j = q - k * cgn;
            const double* Xc = X + (long long)((whole ? c0 : 0) + j) * N;
            const cx<double> a = {Xc[k], k > 0 ? Xc[N - k] : 0.0};
            bufs[(long long)j * L + phys<double, SH>(k)] = cmul(a, pw[k]);
        }
        for (int q = threadIdx.x; q < (L - N) * cg; q += blockDim.x) {
            const int s = q / cg, j = q - s * cg;
            bufs[(long long)j * L + phys<double, SH>(N + s)] = cx<double>{0.0, 0.0};
        }
        __syncthreads();
        blue_convolve<LOG2L>(buf, t, tw, bhat);
        if (cf < cgn) {
            double* xc = X + (long long)((whole ? c0 : 0) + cf) * N;         // this channel's X has been consumed above
            for (int n = t; n < N; n += TEAM) {
                const cx<double> z = buf[phys<double, SH>(n)], w = wconj[n];
                const int time = n < half ? 2 * n : 2 * (N - 1 - n) + 1;
                xc[whole ? time : n] = z.x * w.x - z.y * w.y;                // group mode: store_pcm_group undoes the permutation
            }
        }
        __syncthreads();
        if (!whole) {
            store_pcm_group<-1>(xoff, out, g, f, N, c0, cgn);
            __syncthreads();
        }
    }
    if (whole) store_pcm_f64<-1, false>(xoff, out, g, f, 1, N);
}

namespace {

constexpr size_t kLds = 160 * 1024;
typedef long double ld;

// exp(-i pi p / q)
void unit_ld(long long p, long long q, ld& re, ld& im) {
    const ld PI = 3.14159265358979323846264338327950288419716939937510L;
    long long r = p % (2 * q); if (r < 0) r += 2 * q;
    ld sign = 1.0L;
    if (r >= q) { r -= q; sign = -1.0L; }
    ld c, s;
    if (2 * r <= q) { c = cosl(PI * (ld)r / (ld)q); s = sinl(PI * (ld)r / (ld)q); }
    else { c = -cosl(PI * (ld)(q - r) / (ld)q); s = sinl(PI * (ld)(q - r) / (ld)q); }
    if (r == 0) { c = 1.0L; s = 0.0L; }
    re = sign * c; im = -sign * s;
}

struct BlueTable { cx<double>* wconj = nullptr; cx<double>* bhat = nullptr; cx<double>* pw = nullptr; cx<double>* pw2 = nullptr; int log2l = 0; };
std::mutex g_mu;
std::map<std::pair<int, int>, BlueTable> g_blue;             // (device, N)
thread_local int g_last = 0;
#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last = (int)e_; return FRAD_E_HIP; } } while (0)

int team_for(int log2l) {
    switch (log2l) { case 8: case 9: case 10: return 64; case 11: return 128; case 12: return 256; case 13: return 512; default: return 0; }
}

void host_fft(std::vector<ld>& re, std::vector<ld>& im) {   // in place, forward, radix 2
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t k = 0; k < len / 2; ++k) {
            ld wr, wi; unit_ld(2LL * (long long)k, (long long)len, wr, wi);
            for (size_t i = k; i < n; i += len) {
                const size_t j = i + len / 2;
                const ld tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr; im[i] += ti;
            }
        }
    }
}

int upload(const std::vector<cx<double>>& h, cx<double>** d) {
    BCHK(hipMalloc(d, h.size() * sizeof(cx<double>)));
    BCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
    return FRAD_OK;
}

int get_blue(int N, int log2l, BlueTable& out) {
    int dev = 0; BCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, N);
    auto it = g_blue.find(key);
    if (it != g_blue.end()) { out = it->second; return FRAD_OK; }
    if (g_blue.size() >= 64) {                               // bounded cache of odd frame lengths
        for (auto& kv : g_blue) { (void)hipFree(kv.second.wconj); (void)hipFree(kv.second.bhat); (void)hipFree(kv.second.pw); (void)hipFree(kv.second.pw2); }
        g_blue.clear();
    }
    const int L = 1 << log2l;
    std::vector<cx<double>> wc((size_t)N), pw((size_t)N), pw2((size_t)N), bh((size_t)L);
    std::vector<ld> br((size_t)L, 0.0L), bi((size_t)L, 0.0L);
    for (long long n = 0; n < N; ++n) {
        const long long r = (n * n) % (2LL * N);
        ld re, im; unit_ld(2 * r, 2LL * N, re, im);          // conj(w_n) = exp(-i pi n^2 / N)
        wc[(size_t)n] = cx<double>{(double)re, (double)im};
        br[(size_t)n] = re; bi[(size_t)n] = -im;             // b[n] = w_n, mirrored to the negative lags
        if (n > 0) { br[(size_t)(L - n)] = re; bi[(size_t)(L - n)] = -im; }
        unit_ld(2 * r + n, 2LL * N, re, im);                 // conj(w_k) exp(-i pi k / 2N)
        pw[(size_t)n] = cx<double>{(double)re, (double)im};
        // channel pairs: exp(-i pi k / 2N) w_{N-k} = exp(+i pi (2 (N-k)^2 - k) / 2N); k = 0 pairs with itself (w_0 = 1)
        const long long m = n == 0 ? 0 : (((long long)N - n) * ((long long)N - n)) % (2LL * N);
        unit_ld(n - 2 * m, 2LL * N, re, im);
        pw2[(size_t)n] = cx<double>{(double)re, (double)im};
    }
    host_fft(br, bi);
    for (int k = 0; k < L; ++k) bh[(size_t)k] = cx<double>{(double)(br[(size_t)k] / (ld)L), (double)(bi[(size_t)k] / (ld)L)};
    BlueTable t; t.log2l = log2l;
    int rc = upload(wc, &t.wconj); if (rc != FRAD_OK) return rc;
    rc = upload(bh, &t.bhat); if (rc != FRAD_OK) return rc;
    rc = upload(pw, &t.pw); if (rc != FRAD_OK) return rc;
    rc = upload(pw2, &t.pw2); if (rc != FRAD_OK) return rc;
    g_blue[key] = t; out = t;
    return FRAD_OK;
}

struct BlueCfg { bool ok = false; int log2l = 0, cg = 0, threads = 0, whole = 0, paired = 0; size_t lds = 0; };
BlueCfg blue_cfg(int N, int C, int bits, bool fwd, bool allow_pairs = true) {
    BlueCfg c;
    if (N < 96 || N > 4096) return c;                        // tiny frames: the direct product is cheaper
    if (const char* e = tune("FRAD_TUNE_NO_BLUE")) { if (atoi(e) != 0) return c; }   // A/B knob, not part of the ABI
    int l2 = 8;
    while ((1 << l2) < 2 * N - 1) ++l2;
    if (l2 > 13) return c;
    const int team = team_for(l2);
    const size_t per = (size_t)(1 << l2) * 16, xall = (size_t)C * N * 8;
    long long cg = 0;
    if (xall + per <= kLds) {                                // whole frame: X of all channels next to cg transform buffers
        cg = (long long)((kLds - xall) / per);
        c.whole = 1;
    } else {                                                 // big frame: X of one channel group at a time, per-value I/O
        cg = (long long)(kLds / (per + (size_t)N * 8));
        if (fwd && bits == 12 && (C & 1)) return c;          // packing: 12-bit pairs must not straddle groups (reads may)
    }
    static const bool no_pairs = [] { const char* e = tune("FRAD_TUNE_BLUE_NO_PAIRS"); return e && e[0] == '1'; }();
    // two channels per complex transform -- not for float PCM on the way in: a NaN / Inf sample of one channel would
    // poison its partner through the shared convolution (integers cannot carry one; decode scrubs them first)
    c.paired = (c.whole && (C & 1) == 0 && !no_pairs && allow_pairs) ? 1 : 0;
    const long long need = c.paired ? C / 2 : C;
    if (cg > need) cg = need;
    if (cg * team > 1024) cg = 1024 / team;
    if (!c.whole && fwd && bits == 12) cg &= ~1LL;
    if (cg < 1) return c;
    c.ok = true; c.log2l = l2; c.cg = (int)cg; c.threads = (int)cg * team;
    c.lds = (size_t)cg * per + (c.whole ? xall : (size_t)cg * N * 8);
    return c;
}

template <int LOG2L>
void go_fwd(int lg, const BlueCfg& c, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
            const cx<double>* tw, const BlueTable& t, const Geom& g, int ao) {
#define GO(LGV) do { allow_lds(k_p0_fwd_blue<LOG2L, LGV>, c.lds); \
        hipLaunchKernelGGL((k_p0_fwd_blue<LOG2L, LGV>), grid, dim3(c.threads), c.lds, s, pcm, pay, am, tw, t.wconj, t.bhat, t.pw, t.pw2, g, ao); } while (0)
    switch (lg) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; default: GO(3); break; }
#undef GO
}
template <int LOG2L>
void go_inv(const BlueCfg& c, dim3 grid, hipStream_t s, const unsigned char* pay, double* out, const cx<double>* tw,
            const BlueTable& t, const Geom& g, int ai) {
    allow_lds(k_p0_inv_blue<LOG2L>, c.lds);
    hipLaunchKernelGGL((k_p0_inv_blue<LOG2L>), grid, dim3(c.threads), c.lds, s, pay, out, tw, t.wconj, t.bhat, t.pw, g, ai);
}

int tables_for(int N, int C, int bits, bool fwd, BlueCfg& c, BlueTable& t, const cx<double>** tw, bool allow_pairs = true) {
    c = blue_cfg(N, C, bits, fwd, allow_pairs);
    if (!c.ok) return 0;
    Tables ft; int rc = get_tables(c.log2l, false, ft);
    if (rc != FRAD_OK) return rc;
    rc = get_blue(N, c.log2l, t);
    if (rc != FRAD_OK) return rc;
    *tw = static_cast<const cx<double>*>(ft.tw);
    return 1;
}

}  // namespace

int blue_last_hip_error() { return g_last; }

int blue_prepare(int N) {
    BlueCfg c; BlueTable t; const cx<double>* tw = nullptr;
    const int r = tables_for(N, 1, 32, true, c, t, &tw);
    return r < 0 ? r : FRAD_OK;
}

void blue_clear() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_blue) { (void)hipFree(kv.second.wconj); (void)hipFree(kv.second.bhat); (void)hipFree(kv.second.pw); (void)hipFree(kv.second.pw2); }
    g_blue.clear();
}

// 1: launched, 0: not applicable (caller falls back to the direct kernels), < 0: error
int launch_p0_fwd_blue(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax, Geom g, int aligned_out) {
    if (g.n_frames > 0x7fffffffLL) return 0;
    BlueCfg c; BlueTable t; const cx<double>* tw = nullptr;
    const int r = tables_for(g.N, g.C, g.bits, true, c, t, &tw, (g.dtype >> 3) != 2);
    if (r <= 0) return r;
    if (absmax) BCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)g.n_frames, s));       // atomicMax target
    g.cg = c.cg; g.fpb = 1; g.in_mode = c.whole; g.cc_fast = c.paired;
    dim3 grid((unsigned)g.n_frames);
    switch (c.log2l) {
        case 8: go_fwd<8>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
        case 9: go_fwd<9>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
        case 10: go_fwd<10>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
        case 11: go_fwd<11>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
        case 12: go_fwd<12>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
        default: go_fwd<13>(lg, c, grid, s, pcm, pay, absmax, tw, t, g, aligned_out); break;
    }
    return 1;
}

int launch_p0_inv_blue(hipStream_t s, const unsigned char* pay, double* out, Geom g, int aligned_in) {
    if (g.n_frames > 0x7fffffffLL) return 0;
    BlueCfg c; BlueTable t; const cx<double>* tw = nullptr;
    const int r = tables_for(g.N, g.C, g.bits, false, c, t, &tw);
    if (r <= 0) return r;
    g.cg = c.cg; g.fpb = 1; g.in_mode = c.whole; g.cc_fast = c.paired;
    dim3 grid((unsigned)g.n_frames);
    switch (c.log2l) {
        case 8: go_inv<8>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
        case 9: go_inv<9>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
        case 10: go_inv<10>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
        case 11: go_inv<11>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
        case 12: go_inv<12>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
        default: go_inv<13>(c, grid, s, pay, out, tw, t, g, aligned_in); break;
    }
    return 1;
}

}  // namespace frad
