// profile 1 (psychoacoustic quantiser) entry points -- kernels K7 / K8 (frad_p1.hpp).
#include "frad_p1.hpp"
#include "frad_wave.hpp"
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

#include <climits>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

using namespace frad;

namespace frad {
int p1_digital_out(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C, int32_t bits, int32_t srate,
                   int out_dtype, uint32_t flags, void* out, void* stream);
int launch_p1_fwd_mixed(int lg, hipStream_t s, const unsigned char* pcm, int32_t* q, int32_t* tq, const Geom& g, const P1Tables& tb,
                        int aligned_in, unit_root_fn unit);
int launch_p1_inv_mixed(hipStream_t s, const int32_t* q, const int32_t* tq, double* out, const Geom& g, const P1Tables& tb, unit_root_fn unit);
}

namespace {

// ref: fourier/tools/p1tools.py:4-9 (Hz), fourier/profiles.py:5 (rates), :14-23 (frame sizes)
const long long kEdgesHz[P1_BANDS + 1] = {0, 200, 400, 600, 800, 1000, 1200, 1400, 1600, 2000, 2400, 2800, 3200, 4000, 4800, 5600,
                                          6800, 8000, 9600, 12000, 15600, 20000, 24000, 28800, 34400, 40800, 48000, 4294967295LL};
const int kSrates[12] = {96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000};

bool legal_compact_size(int n) {
    for (int s = 0; s < 8; ++s)
        for (int m : {128, 160, 192, 224}) if (n == (m << s)) return true;
    return false;
}
int valid_srate(int srate) {                     // smallest table rate >= srate (profiles.py:7-8)
    int best = -1;
    for (int r : kSrates) if (r >= srate && (best < 0 || r < best)) best = r;
    return best;
}
int p1_scale_bits(int bits) {                    // profile1.py:16: unknown depth -> 16
    for (int b : {8, 12, 16, 24, 32, 48, 64}) if (b == bits) return bits;
    return 16;
}

thread_local int g_last = 0;

// per-bin band index tables, one per (device, N, snapped rate); a few KiB each, kept for the process lifetime
std::mutex g_band_mu;
std::map<std::tuple<int, int, int>, unsigned char*> g_band;

int band_table(int N, int sr, const P1Tables& tb, const unsigned char** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FRAD_E_HIP;
    std::lock_guard<std::mutex> lk(g_band_mu);
    auto key = std::make_tuple(dev, N, sr);
    auto it = g_band.find(key);
    if (it != g_band.end()) { *out = it->second; return FRAD_OK; }
    std::vector<unsigned char> host((size_t)N, 255);
    for (int b = 0; b < P1_BANDS - 1; ++b) {
        const int a = tb.edge[b] < N ? tb.edge[b] : N, e = tb.edge[b + 1] < N ? tb.edge[b + 1] : N;
        for (int k = a; k < e; ++k) if (host[k] == 255) host[k] = (unsigned char)b;
    }
    unsigned char* d = nullptr;
    if (hipMalloc(&d, (size_t)N) != hipSuccess) return FRAD_E_NOMEM;
    if (hipMemcpy(d, host.data(), (size_t)N, hipMemcpyHostToDevice) != hipSuccess) return FRAD_E_HIP;
    g_band[key] = d; *out = d;
    return FRAD_OK;
}
#define P1CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last = (int)e_; return FRAD_E_HIP; } } while (0)

int make_tables(int N, int srate, int bits, double loss_level, P1Tables& tb) {
    const int sr = valid_srate(srate);
    if (sr < 0) return FRAD_E_INVALID;
    for (int i = 0; i <= P1_BANDS; ++i) {        // Python round(): half-to-even == nearbyint in the default mode
        const double e = nearbyint((double)N / ((double)sr / 2) * (double)kEdgesHz[i]);
        tb.edge[i] = e > (double)INT_MAX ? INT_MAX : (int)e;
    }
    tb.nb_used = P1_BANDS;
    for (int i = 0; i < P1_BANDS; ++i) {
        const int a = tb.edge[i] < N ? tb.edge[i] : N, e = tb.edge[i + 1] < N ? tb.edge[i + 1] : N;
        if (e <= a && tb.nb_used == P1_BANDS) tb.nb_used = i;        // the reference loop breaks at the first empty band
        const double f = (double)(kEdgesHz[i] + kEdgesHz[i + 1]) / 2;
        const double khz = f / 1000.0;
        const double ath = pow(10.0, (3.64 * pow(khz, -0.8) - 6.5 * exp(-0.6 * pow(khz - 3.3, 2.0)) + 1e-3 * pow(khz, 4.0)) / 20);
        tb.floor_[i] = ath < 1.0 ? ath : 1.0;
    }
    tb.scale = ldexp(1.0, p1_scale_bits(bits) - 1);
    tb.loss = fabs(loss_level) > 0.125 ? fabs(loss_level) : 0.125;
    tb.f32 = 0;
    return band_table(N, sr, tb, &tb.band_of);
}

template <int LOG2M>
void go_fwd(int lg, const FastCfg& c, size_t lds, dim3 grid, hipStream_t s, const unsigned char* pcm, int32_t* q, int32_t* tq,
            const Tables& t, const Geom& g, const P1Tables& tb, int ai) {
    const cx<double>* tw = static_cast<const cx<double>*>(t.tw); const cx<double>* post = static_cast<const cx<double>*>(t.post);
#define GO(LGV, MAXT) do { allow_lds(k_p1_fwd<LOG2M, LGV, MAXT>, lds); \
        hipLaunchKernelGGL((k_p1_fwd<LOG2M, LGV, MAXT>), grid, dim3(c.threads), lds, s, pcm, q, tq, tw, post, g, tb, ai); } while (0)
#define GOL(LGV) do { if (c.threads <= 256) GO(LGV, 256); else GO(LGV, 1024); } while (0)
    switch (lg) { case 0: GOL(0); break; case 1: GOL(1); break; case 2: GOL(2); break; default: GOL(3); break; }
#undef GOL
#undef GO
}
template <int LOG2M>
void go_inv(const FastCfg& c, size_t lds, dim3 grid, hipStream_t s, const int32_t* q, const int32_t* tq, double* out,
            const Tables& t, const Geom& g, const P1Tables& tb) {
    const cx<double>* tw = static_cast<const cx<double>*>(t.tw); const cx<double>* post = static_cast<const cx<double>*>(t.post);
    if (c.threads <= 256) { allow_lds(k_p1_inv<LOG2M, 256>, lds); hipLaunchKernelGGL((k_p1_inv<LOG2M, 256>), grid, dim3(c.threads), lds, s, q, tq, out, tw, post, g, tb); }
    else { allow_lds(k_p1_inv<LOG2M, 1024>, lds); hipLaunchKernelGGL((k_p1_inv<LOG2M, 1024>), grid, dim3(c.threads), lds, s, q, tq, out, tw, post, g, tb); }
}

// channel-group variants (frames wider than a CU's LDS): cg channels per pass, <= 512 threads
template <int LOG2M>
void go_fwd_grp(int lg, int threads, size_t lds, dim3 grid, hipStream_t s, const unsigned char* pcm, int32_t* q, int32_t* tq,
                const Tables& t, const Geom& g, const P1Tables& tb) {
    const cx<double>* tw = static_cast<const cx<double>*>(t.tw); const cx<double>* post = static_cast<const cx<double>*>(t.post);
#define GO(LGV) do { allow_lds(k_p1_fwd_grp<LOG2M, LGV, 512>, lds); \
        hipLaunchKernelGGL((k_p1_fwd_grp<LOG2M, LGV, 512>), grid, dim3(threads), lds, s, pcm, q, tq, tw, post, g, tb); } while (0)
    switch (lg) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; default: GO(3); break; }
#undef GO
}
template <int LOG2M>
void go_inv_grp(int threads, size_t lds, dim3 grid, hipStream_t s, const int32_t* q, const int32_t* tq, double* out,
                const Tables& t, const Geom& g, const P1Tables& tb) {
    const cx<double>* tw = static_cast<const cx<double>*>(t.tw); const cx<double>* post = static_cast<const cx<double>*>(t.post);
    allow_lds(k_p1_inv_grp<LOG2M, 512>, lds);
    hipLaunchKernelGGL((k_p1_inv_grp<LOG2M, 512>), grid, dim3(threads), lds, s, q, tq, out, tw, post, g, tb);
}
// group width for profile 1: the FFT buffers plus the quantiser's scratch must fit; 0 = no group geometry
int p1_group(const FastCfg& c, int N, int C, size_t kLdsBytes_, size_t& lds) {
    if (!c.ok || c.cg >= C || c.log2m < 8) return 0;
    const size_t M = (size_t)1 << c.log2m;
    int cg = c.cg;
    while (cg > 0 && ((size_t)cg * M * 16 + p1_scratch_bytes(cg, N) > kLdsBytes_ || cg * c.team > 512)) --cg;
    if (cg < 1) return 0;
    lds = (size_t)cg * M * 16 + p1_scratch_bytes(cg, N);
    return cg;
}

// whole frames per block: shrink the block until the transform buffers plus the quantiser's scratch fit; false = this
// geometry goes to the channel-group, direct or workspace kernels instead
bool p1_fast_fits(FastCfg& c, int N, int C) {
    if (!c.ok || c.cg != C) return false;
    const size_t M = (size_t)1 << c.log2m;
    while (c.fpb > 1 && (size_t)c.fpb * C * M * 16 + p1_scratch_bytes(c.fpb * C, N) > 80 * 1024) c.fpb -= 1;
    c.threads = c.fpb * C * c.team;
    if (c.threads > 1024 || c.threads % 64) return false;
    return (size_t)c.fpb * C * M * 16 + p1_scratch_bytes(c.fpb * C, N) <= 160 * 1024;
}

// |q|^(1/0.75) for q = 0 .. 255 (p1tools.py:44 dequant) and the band thresholds' (e/2)^quant(t), long double -> correctly
// rounded double; one table per device
std::map<int, double*> g_deq;
const double* deq_table() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_band_mu);
    auto it = g_deq.find(dev);
    if (it != g_deq.end()) return it->second;
    double host[768];
    for (int a = 0; a < 256; ++a) host[a] = (double)powl((long double)a, (long double)(1.0 / 0.75));
    // (e/2)^((n + 1/2)^0.75): where the band code round(dequant(log(t) / log(e/2))) of profile1.py:38-40 steps from n to n + 1
    for (int n = 0; n < 256; ++n) host[512 + n] = (double)powl((long double)(2.718281828459045 / 2), powl((long double)n + 0.5L, 0.75L));
    // (e/2)^quant(t), t = 0 .. 255 -- the dequantised band thresholds of profile1.py:63; quant(t) = t^0.75 as in p1w_quant
    for (int t = 0; t < 256; ++t) { const double r = sqrt((double)t), qt = r * sqrt(r); host[256 + t] = (double)powl((long double)(2.718281828459045 / 2), (long double)qt); }
    double* d = nullptr;
    if (hipMalloc(&d, sizeof host) != hipSuccess) return nullptr;
    if (hipMemcpy(d, host, sizeof host, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    g_deq[dev] = d;
    return d;
}

P1Wave wave_tables(const P1Tables& tb, int N) {
    P1Wave pw{};
    pw.deq = nullptr;
    for (int i = 0; i < 28; ++i) pw.edge[i] = tb.edge[i] < N ? tb.edge[i] : N;
    for (int i = 0; i < 27; ++i) pw.floor_[i] = tb.floor_[i];
    pw.scale = tb.scale; pw.loss = tb.loss; pw.nb_used = tb.nb_used; pw.band_of = tb.band_of; pw.tq_in = nullptr; pw.tq_out = nullptr;
    return pw;
}
// the root-of-unity generator the wave table blob is built with (the same values as frad_hip.hip's tables)
void p1_unit_neg(long long p, long long q, long double& re, long double& im) {
    const long double PI = 3.14159265358979323846264338327950288419716939937510L;
    long long r = p % (2 * q); if (r < 0) r += 2 * q;
    const long long h = q / 2;
    const int quad = (int)(r / h);
    const long long rem = r % h;
    long double c, s;
    if (4 * rem <= q) { c = cosl(PI * (long double)rem / (long double)q); s = sinl(PI * (long double)rem / (long double)q); }
    else { c = sinl(PI * (long double)(h - rem) / (long double)q); s = cosl(PI * (long double)(h - rem) / (long double)q); }
    if (rem == 0) { c = 1.0L; s = 0.0L; }
    long double C, S;
    switch (quad) { case 0: C = c; S = s; break; case 1: C = -s; S = c; break; case 2: C = -c; S = -s; break; default: C = s; S = -c; }
    re = C; im = -S;
}

Geom p1_geom(long long n_frames, int N, int C, long long stride, int n_valid, int dtype, uint32_t flags) {
    Geom g{};
    g.n_frames = n_frames; g.frame_stride = stride; g.payload_stride = 0; g.N = N; g.C = C; g.bits = 32; g.le = 0;
    g.dtype = dtype; g.raw_be = (flags & FRAD_RAW_BE_INTS) ? 1 : 0; g.fpb = 1; g.n_valid = n_valid; g.cg = C; g.in_mode = 0; g.cc_fast = 0;
    return g;
}
constexpr size_t kLds = 160 * 1024;

}  // namespace

namespace frad {
void p1_clear() {                                            // frad_plan_clear: the per-(N, rate) band maps
    std::lock_guard<std::mutex> lk(g_band_mu);
    for (auto& kv : g_band) (void)hipFree(kv.second);
    g_band.clear();
    for (auto& kv : g_deq) (void)hipFree(kv.second);
    g_deq.clear();
}
}  // namespace frad

extern "C" {

int frad_p1_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                     int32_t n_valid, int32_t bits, int32_t srate, double loss_level, uint32_t flags,
                     int32_t* q, int32_t* tq, void* stream) {
    if (n_frames < 0 || C < 1 || C > 64 || !legal_compact_size(N) || n_valid < 0 || n_valid > N) return FRAD_E_INVALID;
    if (pcm_dtype < 0 || pcm_dtype > 23) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!pcm || !q || !tq) return FRAD_E_INVALID;
    const int kind = pcm_dtype >> 3, lg = (pcm_dtype >> 1) & 3;
    if ((kind == 2 && lg == 0) || (lg == 0 && (pcm_dtype & 1))) return FRAD_E_INVALID;
    P1Tables tb;
    int rc = make_tables(N, srate, bits, loss_level, tb);
    if (rc != FRAD_OK) return rc;
    // f32 / f16 PCM is not widened by the reference (pcmformat.py:35): float32 DCT, float32 band statistics, float64
    // quantiser.  Here the DCT runs in float64 on the exactly widened samples and its result is rounded to float32.
    tb.f32 = (kind == 2 && lg <= 2) ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Geom g = p1_geom(n_frames, N, C, frame_stride, n_valid, pcm_dtype, flags);
    const int ai = ((reinterpret_cast<uintptr_t>(pcm) & 15u) == 0 && (((frame_stride * C) << lg) % 16 == 0) &&
                    ((((long long)N * C) << lg) % 16 == 0)) ? 1 : 0;
    const unsigned char* in = static_cast<const unsigned char*>(pcm);
    if (!tb.f32) {
        P1Wave pw = wave_tables(tb, N);
        pw.tq_out = tq;
        const double* dt = deq_table();
        pw.tqh = dt ? dt + 512 : nullptr;
        if (launch_p1_fwd_wave(lg, s, in, q, g, pw, ai, p1_unit_neg)) { P1CHK(hipGetLastError()); return FRAD_OK; }
    }
    FastCfg c = fast_cfg(N, C, false);
    if (p1_fast_fits(c, N, C)) {
        const int M = 1 << c.log2m;
        const size_t lds = (size_t)c.fpb * C * M * 16 + p1_scratch_bytes(c.fpb * C, N);
        Tables t; rc = get_tables(c.log2m, false, t);
        if (rc != FRAD_OK) return rc;
        g.fpb = c.fpb;
        dim3 grid((unsigned)((n_frames + c.fpb - 1) / c.fpb));
        switch (c.log2m) {
            case 6: go_fwd<6>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 7: go_fwd<7>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 8: go_fwd<8>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 9: go_fwd<9>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 10: go_fwd<10>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 11: go_fwd<11>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            case 12: go_fwd<12>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
            default: go_fwd<13>(lg, c, lds, grid, s, in, q, tq, t, g, tb, ai); break;
        }
    } else if (size_t glds = 0; int cgw = p1_group(c, N, C, kLds, glds)) {
        if (n_frames > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
        Tables t; rc = get_tables(c.log2m, false, t);
        if (rc != FRAD_OK) return rc;
        g.fpb = 1; g.cg = cgw;
        dim3 grid((unsigned)n_frames);
        const int threads = cgw * c.team;
        switch (c.log2m) {
            case 8: go_fwd_grp<8>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
            case 9: go_fwd_grp<9>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
            case 10: go_fwd_grp<10>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
            case 11: go_fwd_grp<11>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
            case 12: go_fwd_grp<12>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
            default: go_fwd_grp<13>(lg, threads, glds, grid, s, in, q, tq, t, g, tb); break;
        }
    } else {
        const size_t lds = 2 * (size_t)N * C * 8 + p1_scratch_bytes(C, N);
        if (n_frames > 0x7fffffffLL / C) return FRAD_E_UNSUPPORTED;
        {                                                    // {160, 192, 224} x 2^n: mixed-radix FFT (frad_mixed.hip)
            const int r = launch_p1_fwd_mixed(lg, s, in, q, tq, g, tb, ai, p1_unit_neg);
            if (r < 0) { if (r == FRAD_E_HIP) g_last = mixed_last_hip_error(); return r; }
            if (r == 1) return FRAD_OK;
        }
        if (lds > kLds) {
            rc = global_p1_analogue(in, q, tq, g, tb, s);
            if (rc == FRAD_E_HIP) g_last = global_last_hip_error();
            return rc;
        }
        DirectTable d; rc = get_direct(N, d);
        if (rc != FRAD_OK) return rc;
        dim3 grid((unsigned)n_frames);
#define GOD(LGV) do { allow_lds(k_p1_fwd_direct<LGV>, lds); hipLaunchKernelGGL((k_p1_fwd_direct<LGV>), grid, dim3(256), lds, s, in, q, tq, d.ct, g, tb, ai); } while (0)
        switch (lg) { case 0: GOD(0); break; case 1: GOD(1); break; case 2: GOD(2); break; default: GOD(3); break; }
#undef GOD
    }
    P1CHK(hipGetLastError());
    return FRAD_OK;
}

int frad_p1_digital(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C, int32_t bits, int32_t srate,
                    double* pcm_out, void* stream) {
    return p1_digital_out(q, tq, n_frames, N, C, bits, srate, FRAD_PCM_F64LE, 0, pcm_out, stream);
}

}  // extern "C"
namespace frad {
// frad_p1_digital, optionally with the decoder's output conversion applied by the kernel's own store (frad_p1_digital_pcm):
// FRAD_OK = done; 1 = this geometry's kernel cannot convert (N = 2048 wave kernel, frames wider than a CU): second pass
int p1_digital_out(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C, int32_t bits, int32_t srate,
                   int out_dtype, uint32_t flags, void* out, void* stream) {
    const bool conv = out_dtype != FRAD_PCM_F64LE;
    double* pcm_out = static_cast<double*>(out);
    if (n_frames < 0 || C < 1 || C > 64 || !legal_compact_size(N)) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!q || !tq || !pcm_out) return FRAD_E_INVALID;
    P1Tables tb;
    int rc = make_tables(N, srate, bits, 1.0, tb);
    if (rc != FRAD_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Geom g = p1_geom(n_frames, N, C, N, N, conv ? out_dtype : FRAD_PCM_F64LE, conv ? flags : 0);
    FastCfg c = fast_cfg(N, C, false);
    if (conv && N == 2048 && C <= 2) return 1;                  // (the table-driven wave kernel writes float64)
    if (N == 2048 && C <= 2 && n_frames < 0x7fffffffLL && p1_fast_fits(c, N, C)) {
        // N = 2048, one or two channels: the table-driven wave kernel, then the exact kernel over the frames it marked (a band
        // code outside [0, 256) or |q| >= 256: none in a sane stream).  The list is stream-ordered scratch: count + one slot per frame.
        struct Redo { hipStream_t s; int* p = nullptr; ~Redo() { if (p) (void)hipFreeAsync(p, s); } } redo{s};
        if (hipMallocAsync(reinterpret_cast<void**>(&redo.p), sizeof(int) * (size_t)(n_frames + 1), s) != hipSuccess) return FRAD_E_NOMEM;
        P1CHK(hipMemsetAsync(redo.p, 0, sizeof(int), s));
        P1Wave pw = wave_tables(tb, N);
        pw.tq_in = tq;
        pw.deq = deq_table();
        pw.redo = redo.p;
        if (launch_p1_inv_wave(s, q, pcm_out, g, pw, p1_unit_neg)) {
            P1CHK(hipGetLastError());
            const int M = 1 << c.log2m;
            const size_t lds = (size_t)C * M * 16 + p1_scratch_bytes(C, N);
            Tables t; rc = get_tables(c.log2m, false, t);
            if (rc != FRAD_OK) return rc;
            Geom gr = g; gr.fpb = 1;
            const int threads = C * c.team;
            const cx<double>* tw = static_cast<const cx<double>*>(t.tw); const cx<double>* post = static_cast<const cx<double>*>(t.post);
            const unsigned grid = (unsigned)(n_frames < 256 ? n_frames : 256);
            allow_lds(k_p1_inv_redo<10, 256>, lds);
            hipLaunchKernelGGL((k_p1_inv_redo<10, 256>), dim3(grid), dim3(threads), lds, s, q, tq, pcm_out, tw, post, gr, tb, redo.p);
            P1CHK(hipGetLastError());
            return FRAD_OK;
        }
    }
    if (p1_fast_fits(c, N, C)) {
        const int M = 1 << c.log2m;
        const size_t lds = (size_t)c.fpb * C * M * 16 + p1_scratch_bytes(c.fpb * C, N);
        Tables t; rc = get_tables(c.log2m, false, t);
        if (rc != FRAD_OK) return rc;
        g.fpb = c.fpb;
        dim3 grid((unsigned)((n_frames + c.fpb - 1) / c.fpb));
        switch (c.log2m) {
            case 6: go_inv<6>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 7: go_inv<7>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 8: go_inv<8>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 9: go_inv<9>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 10: go_inv<10>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 11: go_inv<11>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 12: go_inv<12>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
            default: go_inv<13>(c, lds, grid, s, q, tq, pcm_out, t, g, tb); break;
        }
    } else if (size_t glds = 0; int cgw = p1_group(c, N, C, kLds, glds)) {
        if (n_frames > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
        Tables t; rc = get_tables(c.log2m, false, t);
        if (rc != FRAD_OK) return rc;
        g.fpb = 1; g.cg = cgw;
        dim3 grid((unsigned)n_frames);
        const int threads = cgw * c.team;
        switch (c.log2m) {
            case 8: go_inv_grp<8>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 9: go_inv_grp<9>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 10: go_inv_grp<10>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 11: go_inv_grp<11>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
            case 12: go_inv_grp<12>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
            default: go_inv_grp<13>(threads, glds, grid, s, q, tq, pcm_out, t, g, tb); break;
        }
    } else {
        const size_t lds = 2 * (size_t)N * C * 8 + p1_scratch_bytes(C, N);
        if (n_frames > 0x7fffffffLL / C) return FRAD_E_UNSUPPORTED;
        {
            const int r = launch_p1_inv_mixed(s, q, tq, pcm_out, g, tb, p1_unit_neg);
            if (r < 0) { if (r == FRAD_E_HIP) g_last = mixed_last_hip_error(); return r; }
            if (r == 1) return FRAD_OK;
        }
        if (lds > kLds) {
            if (conv) return 1;
            rc = global_p1_digital(q, tq, pcm_out, g, tb, s);
            if (rc == FRAD_E_HIP) g_last = global_last_hip_error();
            return rc;
        }
        DirectTable d; rc = get_direct(N, d);
        if (rc != FRAD_OK) return rc;
        allow_lds(k_p1_inv_direct<0>, lds);
        hipLaunchKernelGGL(k_p1_inv_direct<0>, dim3((unsigned)n_frames), dim3(256), lds, s, q, tq, pcm_out, d.ct, g, tb);
    }
    P1CHK(hipGetLastError());
    return FRAD_OK;
}

}  // namespace frad
extern "C" {

int frad_p1_overlap_add(const double* frames, int64_t n_frames, int32_t N, int32_t C, int32_t overlap_ratio,
                        const double* prev_tail, double* ola_out, double* next_tail, void* stream) {
    if (n_frames < 0 || N < 1 || C < 1 || overlap_ratio < 2 || overlap_ratio > 256) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!frames || !ola_out) return FRAD_E_INVALID;
    const int cut = (int)((long long)N * (overlap_ratio - 1) / overlap_ratio);     // decoder.py:44
    const long long total = n_frames * (long long)cut * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_p1_ola<0>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), frames, (long long)n_frames, N, C,
                       cut, prev_tail, ola_out, next_tail);
    P1CHK(hipGetLastError());
    return FRAD_OK;
}

}  // extern "C"
