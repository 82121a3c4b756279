// frad_hip.hip -- C-ABI entry points of libfrad_hip.so (see include/frad_hip.h) and kernel launch
// logic.  Built with  hipcc --offload-arch=gfx950 -shared -fPIC  (see __graft_entry__.build()).
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

using namespace frad;

namespace {

thread_local int g_last_hip = 0;
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_hip = (int)e_; return FRAD_E_HIP; } } while (0)

constexpr int kLdsBytes = 160 * 1024;     // CDNA4: 160 KiB LDS per CU, one workgroup may own it all

bool valid_bits(int b) { return b == 12 || b == 16 || b == 24 || b == 32 || b == 48 || b == 64; }
bool valid_dtype(int d) {
    if (d < 0 || d > 23) return false;
    const int kind = d >> 3, lg = (d >> 1) & 3, be = d & 1;
    if (kind == 2 && lg == 0) return false;
    if (lg == 0 && be) return false;
    return true;
}
int log2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
int gcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// exp(-i * pi * p / q) for even q, exact at the multiples of pi/2 and symmetric inside octants.
void unit_neg(long long p, long long q, long double& re, long double& im) {
    const long double PI = 3.14159265358979323846264338327950288419716939937510L;
    long long r = p % (2 * q); if (r < 0) r += 2 * q;
    const long long h = q / 2;                 // q is even
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


std::mutex g_mu;
std::map<std::tuple<int, int, int>, Tables> g_tables;        // (device, log2M, f32)
std::map<std::pair<int, int>, DirectTable> g_direct;         // (device, N)

template <typename T>
int build_tables(int log2m, Tables& out) {
    const int M = 1 << log2m, N = 2 * M;
    std::vector<cx<T>> tw(M), post(2 * (M / 2 + 1));
    for (int k = 0; k < M; ++k) {
        long double re, im; unit_neg(2LL * k, M, re, im);
        tw[k].x = (T)re; tw[k].y = (T)im;
    }
    for (int k = 0; k <= M / 2; ++k) {
        long double re, im;
        unit_neg(k, 2LL * N, re, im);              // w_k = exp(-i pi k / 2N)
        post[2 * k].x = (T)re; post[2 * k].y = (T)im;
        unit_neg((long long)N + 5LL * k, 2LL * N, re, im);   // g_k = exp(-i pi (1/2 + 5k/2N))
        post[2 * k + 1].x = (T)re; post[2 * k + 1].y = (T)im;
    }
    HIPCHK(hipMalloc(&out.tw, tw.size() * sizeof(cx<T>)));
    HIPCHK(hipMalloc(&out.post, post.size() * sizeof(cx<T>)));
    HIPCHK(hipMemcpy(out.tw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.post, post.data(), post.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    std::vector<unsigned char> blob;
    for (int which = 0; which < 3; ++which) {                 // plan A, plan B, inverse plan I
        if (!pers_blob_build(log2m, sizeof(T) == 4, which, blob, unit_neg)) continue;
        void*& dst = which == 0 ? out.blob : which == 1 ? out.blob_b : out.blob_i;
        HIPCHK(hipMalloc(&dst, blob.size()));
        HIPCHK(hipMemcpy(dst, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    return FRAD_OK;
}

}  // namespace
namespace frad {
void unit_root(long long p, long long q, long double& re, long double& im) { unit_neg(p, q, re, im); }
int get_tables(int log2m, bool f32, Tables& out) {
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, log2m, (int)f32);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { out = it->second; return FRAD_OK; }
    Tables t;
    const int rc = f32 ? build_tables<float>(log2m, t) : build_tables<double>(log2m, t);
    if (rc != FRAD_OK) return rc;
    g_tables[key] = t; out = t;
    return FRAD_OK;
}

int get_direct(int N, DirectTable& out) {
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, N);
    auto it = g_direct.find(key);
    if (it != g_direct.end()) { out = it->second; return FRAD_OK; }
    if (g_direct.size() >= 64) {                       // bounded cache of odd frame lengths
        for (auto& kv : g_direct) (void)hipFree(kv.second.ct);
        g_direct.clear();
    }
    std::vector<double> ct(4 * (size_t)N);
    for (long long j = 0; j < 4LL * N; ++j) { long double re, im; unit_neg(j, 2LL * N, re, im); ct[j] = (double)re; }
    DirectTable t;
    HIPCHK(hipMalloc(&t.ct, ct.size() * sizeof(double)));
    HIPCHK(hipMemcpy(t.ct, ct.data(), ct.size() * sizeof(double), hipMemcpyHostToDevice));
    g_direct[key] = t; out = t;
    return FRAD_OK;
}

// launch geometry of the FFT kernels
int team_of(int log2m) {
    switch (log2m) { case 6: return 16; case 7: return 32; case 8: case 9: case 10: return 64;
                     case 11: return 128; case 12: return 256; case 13: return 512; default: return 0; }
}
FastCfg fast_cfg(int N, int C, bool f32) {
    FastCfg c;
    const int l2 = log2_exact(N);
    if (l2 < 7 || l2 > 14 || C < 1) return c;
    c.log2m = l2 - 1; c.team = team_of(c.log2m);
    const int M = 1 << c.log2m;
    const size_t per_cf = (size_t)padded_slots(M) * (f32 ? 8 : 16);
    const long long pft = (long long)C * c.team;
    const size_t pfl = per_cf * (size_t)C;
    int q = 1;
    if (c.team < 64) { const int w = 64 / c.team; q = w / gcd(C, w); }
    if (q * pft > 1024 || q * pfl > (size_t)kLdsBytes) {
        // one frame's channels exceed a CU: transform `cg` channels per pass (per-value I/O)
        if (c.team < 64) return c;
        int cg = C;
        // <= 512 threads for float64: two waves per SIMD leave the radix-16 butterflies their 256 registers
        const long long tmax = f32 ? 1024 : 512;
        while (cg > 1 && ((long long)cg * c.team > tmax || (size_t)cg * per_cf > (size_t)kLdsBytes)) --cg;
        if ((long long)cg * c.team > 1024 || (size_t)cg * per_cf > (size_t)kLdsBytes) return c;
        if (cg > 1 && (cg & 1)) --cg;                       // even groups keep 12-bit pairs together
        if (const char* e = tune("FRAD_TUNE_CG")) { const int v = atoi(e); if (v >= 1 && v <= cg && (v == 1 || !(v & 1))) cg = v; }
        c.cg = cg; c.fpb = 1; c.threads = cg * c.team; c.lds = (size_t)cg * per_cf; c.ok = true;
        return c;
    }
    int fpb = q;
    // 256-thread blocks keep the whole register file available to the radix-16 butterflies; two
    // such blocks (<= 80 KiB of LDS each) share a CU so that one streams while the other computes
    while ((fpb + q) * pft <= 256 && (size_t)(fpb + q) * pfl <= 80 * 1024) fpb += q;
    c.cg = C; c.fpb = fpb; c.threads = (int)(fpb * pft); c.lds = fpb * pfl; c.ok = true;
    if (const char* e = tune("FRAD_TUNE_FPB")) {          // tuning knobs for experiments (not part of the ABI)
        const int v = atoi(e);
        if (v >= q && v % q == 0 && v * pft <= 1024 && (size_t)v * pfl <= (size_t)kLdsBytes) { c.fpb = v; c.threads = (int)(v * pft); c.lds = v * pfl; }
    }
    if (const char* e = tune("FRAD_TUNE_LDS_PAD")) c.lds += (size_t)atoi(e);
    return c;
}

}  // namespace frad
namespace {

int check_common(const void* a, const void* b, long long n_frames, int N, int C, int bits) {
    if (n_frames < 0 || N < 1 || C < 1 || C > 256 || !valid_bits(bits)) return FRAD_E_INVALID;
    if (n_frames > 0 && (a == nullptr || b == nullptr)) return FRAD_E_INVALID;
    if ((long long)N * C > (1LL << 30)) return FRAD_E_UNSUPPORTED;
    return FRAD_OK;
}

// direct (any-N) kernels: frames per block so that a thread's table fetch feeds up to 8 channel-frames, but
// never so many that CUs stay idle or the staging buffers outgrow the LDS; one thread per output index
int direct_fpb(long long n_frames, int C, size_t per_frame_lds) {
    long long fpb = C >= 8 ? 1 : 8 / C;
    const long long spread = n_frames / 256;                  // keep >= one block per CU when the batch allows
    if (fpb > spread) fpb = spread < 1 ? 1 : spread;
    while (fpb > 1 && (size_t)fpb * per_frame_lds > (size_t)kLdsBytes) --fpb;
    return (int)fpb;
}
int direct_threads(int N) {
    const int t = ((N + 63) / 64) * 64;
    return t > 1024 ? 1024 : t;
}

int blocks_per_frame(long long work_items) {
    long long b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : b > 64 ? 64 : b);
}

// ---- kernel selection by template expansion -------------------------------------------------
template <int BITS>
int launch_p4_pack_lg(int lg, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am, const Geom& g, int bpf) {
    switch (lg) {
        case 0: hipLaunchKernelGGL((k_p4_pack<BITS, 0>), grid, dim3(256), 0, s, pcm, pay, am, g, bpf); break;
        case 1: hipLaunchKernelGGL((k_p4_pack<BITS, 1>), grid, dim3(256), 0, s, pcm, pay, am, g, bpf); break;
        case 2: hipLaunchKernelGGL((k_p4_pack<BITS, 2>), grid, dim3(256), 0, s, pcm, pay, am, g, bpf); break;
        default: hipLaunchKernelGGL((k_p4_pack<BITS, 3>), grid, dim3(256), 0, s, pcm, pay, am, g, bpf); break;
    }
    return FRAD_OK;
}

int launch_p0_fwd_f64(int lg, const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                      double* am, const Tables& tb, const Geom& g, int ai, int ao) {
    return c.log2m <= 9 ? launch_p0_fwd_f64_lo(lg, c, grid, s, pcm, pay, am, tb, g, ai, ao)
                        : launch_p0_fwd_f64_hi(lg, c, grid, s, pcm, pay, am, tb, g, ai, ao);
}

Geom make_geom(long long n_frames, int N, int C, long long frame_stride, long long payload_stride, int bits, uint32_t flags, int dtype) {
    Geom g{};
    g.n_frames = n_frames; g.frame_stride = frame_stride; g.payload_stride = payload_stride;
    g.N = N; g.C = C; g.bits = bits; g.le = (flags & FRAD_LITTLE_ENDIAN) ? 1 : 0; g.dtype = dtype;
    g.raw_be = (flags & FRAD_RAW_BE_INTS) ? 1 : 0; g.fpb = 1; g.n_valid = N; g.cg = C; g.in_mode = 0; g.cc_fast = 0;
    g.ovf_flag = nullptr; g.ovf_limit = 0.0;
    return g;
}

int p0_analogue_impl(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                     int32_t bits, uint32_t flags, void* payload, int64_t payload_stride, double* absmax,
                     int32_t* overflow_flag, void* stream, int fpc, long long clip_stride);

bool input_aligned(const void* pcm, long long frame_stride, int N, int C, int lg) {
    return aligned16(pcm) && (((frame_stride * C) << lg) % 16 == 0) && ((((long long)N * C) << lg) % 16 == 0);
}

}  // namespace

extern "C" {

int frad_abi_version(void) { return FRAD_ABI_VERSION; }
int frad_last_hip_error(void) { return g_last_hip; }

const char* frad_strerror(int status) {
    switch (status) {
        case FRAD_OK: return "ok";
        case FRAD_E_INVALID: return "invalid argument";
        case FRAD_E_UNSUPPORTED: return "geometry not supported by the HIP transform core (more than 2^30 values per frame, or a batch beyond the grid limit)";
        case FRAD_E_HIP: return "HIP runtime error (no MI355X visible, or a launch failed); see frad_last_hip_error()";
        case FRAD_E_NOMEM: return "out of device memory";
        default: return "unknown frad_status";
    }
}

size_t frad_payload_bytes(int32_t N, int32_t C, int32_t bits) {
    const size_t n = (size_t)N * (size_t)C;
    return bits == 12 ? (n * 3 + 1) / 2 : n * (size_t)bits / 8;
}

int frad_has_fast_path(int32_t N, int32_t C, int32_t compute_f32) { return fast_cfg(N, C, compute_f32 != 0).ok ? 1 : 0; }

int frad_plan_prepare(int32_t N, int32_t compute_f32) {
    const int l2 = log2_exact(N);
    if (l2 >= 7 && l2 <= 14) { Tables t; return get_tables(l2 - 1, compute_f32 != 0, t); }
    if (N < 1) return FRAD_E_INVALID;
    { const int rc = mixed_prepare(N, unit_neg); if (rc != FRAD_OK) return rc; }
    if (!compute_f32) { const int rc = blue_prepare(N); if (rc != FRAD_OK) return rc; }
    DirectTable d; return get_direct(N, d);
}

void frad_plan_clear(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_tables) { (void)hipFree(kv.second.tw); (void)hipFree(kv.second.post); if (kv.second.blob) (void)hipFree(kv.second.blob); if (kv.second.blob_b) (void)hipFree(kv.second.blob_b); if (kv.second.blob_i) (void)hipFree(kv.second.blob_i); }
    for (auto& kv : g_direct) (void)hipFree(kv.second.ct);
    g_tables.clear(); g_direct.clear();
    blue_clear();
    mixed_clear();
    crc_clear();
    p1_clear();
    wave_clear();
}

int frad_p4_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                     int32_t bits, uint32_t flags, void* payload, int64_t payload_stride, double* absmax, void* stream) {
    int rc = check_common(pcm, payload, n_frames, N, C, bits);
    if (rc != FRAD_OK) return rc;
    if (!valid_dtype(pcm_dtype)) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (payload_stride < (int64_t)frad_payload_bytes(N, C, bits)) return FRAD_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (absmax) HIPCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)n_frames, s));
    const int lg = (pcm_dtype >> 1) & 3;
    Geom g = make_geom(n_frames, N, C, frame_stride, payload_stride, bits, flags, pcm_dtype);
    const long long NC = (long long)N * C;
    const int U = unit_values(bits);
    const bool fast = NC >= U && aligned16(pcm) && ((frame_stride * C) << lg) % 16 == 0 && aligned16(payload) && payload_stride % 16 == 0;
    int bpf = blocks_per_frame(fast ? (NC / U + 3) / 4 : NC);                // fast path: four units per thread
    if (n_frames * bpf > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
    dim3 grid((unsigned)(n_frames * bpf));
    // small frames (fewer than four units per thread of one block): a wave per frame, four frames per block (bpf = 0)
    static const bool no_wave = tune("FRAD_TUNE_NO_P4_WAVE") != nullptr;
    if (fast && !no_wave && NC / U < 1024) { bpf = 0; grid = dim3((unsigned)((n_frames + 3) / 4)); }
    const unsigned char* in = static_cast<const unsigned char*>(pcm);
    unsigned char* out = static_cast<unsigned char*>(payload);
    if (fast) {
        switch (bits) {
            case 12: launch_p4_pack_lg<12>(lg, grid, s, in, out, absmax, g, bpf); break;
            case 16: launch_p4_pack_lg<16>(lg, grid, s, in, out, absmax, g, bpf); break;
            case 24: launch_p4_pack_lg<24>(lg, grid, s, in, out, absmax, g, bpf); break;
            case 32: launch_p4_pack_lg<32>(lg, grid, s, in, out, absmax, g, bpf); break;
            case 48: launch_p4_pack_lg<48>(lg, grid, s, in, out, absmax, g, bpf); break;
            default: launch_p4_pack_lg<64>(lg, grid, s, in, out, absmax, g, bpf); break;
        }
    } else {
        hipLaunchKernelGGL(k_p4_pack_slow<0>, grid, dim3(256), 0, s, in, out, absmax, g, bpf);
    }
    HIPCHK(hipGetLastError());
    return FRAD_OK;
}

int frad_p4_digital(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                    uint32_t flags, double* pcm_out, void* stream) {
    int rc = check_common(payload, pcm_out, n_frames, N, C, bits);
    if (rc != FRAD_OK) return rc;
    if (n_frames == 0) return FRAD_OK;
    if (payload_stride < (int64_t)frad_payload_bytes(N, C, bits)) return FRAD_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Geom g = make_geom(n_frames, N, C, N, payload_stride, bits, flags, FRAD_PCM_F64LE);
    const long long NC = (long long)N * C;
    const int U = unit_values(bits);
    const bool fast = NC >= U && aligned16(payload) && payload_stride % 16 == 0 && aligned16(pcm_out) && (NC % 2 == 0);
    // a pair of values per thread keeps the float64 stores contiguous: whole 1 KiB rows per store instruction, nontemporal
    // (16 bit 4.0 -> 7.1 TB/s; 12 / 24 / 32 bit 2.8 / 5.5 / 5.9 -> 7.0 TB/s in round 3; `fast`: NC even, aligned rows)
    static const bool pairs24 = tune("FRAD_TUNE_NO_P4_PAIRS24") == nullptr;
    static const bool pairs32 = tune("FRAD_TUNE_NO_P4_PAIRS32") == nullptr;
    const bool pairs = fast && (bits == 16 || bits == 12 || (bits == 24 && pairs24) || (bits == 32 && pairs32)) && !tune("FRAD_TUNE_NO_P4_PAIRS");   // (12 / 24 bit: 3 / 6 payload bytes per pair)
    const bool sub12 = fast && !pairs && (bits == 24 || bits == 48) && !tune("FRAD_TUNE_NO_P4_PAIRS");   // same idea, 12-byte sub-units
    const int bpf = blocks_per_frame(pairs ? (NC / 2 + 3) / 4 : sub12 ? (NC / (96 / bits) + 3) / 4 : fast ? NC / U : NC);
    if (n_frames * bpf > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
    dim3 grid((unsigned)(n_frames * bpf));
    const unsigned char* in = static_cast<const unsigned char*>(payload);
    if (pairs) {
        if (bits == 12) hipLaunchKernelGGL(k_p4_unpack_3b<12>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
        else if (bits == 24) hipLaunchKernelGGL(k_p4_unpack_3b<24>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
        else if (bits == 32) hipLaunchKernelGGL(k_p4_unpack_pairs<32>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
        else hipLaunchKernelGGL(k_p4_unpack_pairs<16>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
    } else if (sub12) {
        if (bits == 24) hipLaunchKernelGGL(k_p4_unpack_12b<24>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
        else hipLaunchKernelGGL(k_p4_unpack_12b<48>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
    } else if (fast) {
        switch (bits) {
            case 12: hipLaunchKernelGGL(k_p4_unpack<12>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
            case 16: hipLaunchKernelGGL(k_p4_unpack<16>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
            case 24: hipLaunchKernelGGL(k_p4_unpack<24>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
            case 32: hipLaunchKernelGGL(k_p4_unpack<32>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
            case 48: hipLaunchKernelGGL(k_p4_unpack<48>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
            default: hipLaunchKernelGGL(k_p4_unpack<64>, grid, dim3(256), 0, s, in, pcm_out, g, bpf); break;
        }
    } else {
        hipLaunchKernelGGL(k_p4_unpack_slow<0>, grid, dim3(256), 0, s, in, pcm_out, g, bpf);
    }
    HIPCHK(hipGetLastError());
    return FRAD_OK;
}

static double storage_float_max(int bits) {                  // FLOAT_DR, profile0.py:6-13
    return bits <= 16 ? 65504.0 : bits <= 32 ? 3.4028234663852886e38 : 1.7976931348623157e308;
}

int frad_p0_overflow_scan(const double* absmax, int64_t n_frames, int32_t bits, int32_t* flag, void* stream) {
    if (n_frames < 0 || !valid_bits(bits)) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!absmax || !flag) return FRAD_E_INVALID;
    const double lim = storage_float_max(bits);
    long long blocks = (n_frames + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_overflow_scan<0>, dim3((unsigned)blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), absmax, (long long)n_frames, lim, flag);
    HIPCHK(hipGetLastError());
    return FRAD_OK;
}

int frad_p0_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                     int32_t bits, uint32_t flags, void* payload, int64_t payload_stride, double* absmax, void* stream) {
    return frad_p0_analogue_checked(pcm, pcm_dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, nullptr, stream);
}

int frad_p0_analogue_checked(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                             int32_t bits, uint32_t flags, void* payload, int64_t payload_stride, double* absmax,
                             int32_t* overflow_flag, void* stream) {
    return p0_analogue_impl(pcm, pcm_dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, overflow_flag, stream, 0, 0);
}

int frad_p0_analogue_clips(const void* pcm, int32_t pcm_dtype, int64_t n_clips, int64_t clip_stride, int32_t frames_per_clip,
                           int32_t N, int32_t C, int32_t bits, uint32_t flags, void* payload, int64_t payload_stride,
                           double* absmax, int32_t* overflow_flag, void* stream) {
    if (n_clips < 0 || frames_per_clip < 1 || N < 1 || clip_stride < (int64_t)frames_per_clip * N) return FRAD_E_INVALID;
    if (n_clips > 0x7fffffffLL / frames_per_clip) return FRAD_E_UNSUPPORTED;
    return p0_analogue_impl(pcm, pcm_dtype, n_clips * frames_per_clip, N, C, N, bits, flags, payload, payload_stride, absmax, overflow_flag,
                            stream, frames_per_clip, clip_stride);
}

}  // extern "C"
namespace {

// frames of the clips gathered into / scattered from a dense stream-ordered scratch (the kernels without clip addressing)
struct StreamScratch { hipStream_t s; void* p = nullptr; ~StreamScratch() { if (p) (void)hipFreeAsync(p, s); } };

int p0_analogue_impl(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C, int64_t frame_stride,
                     int32_t bits, uint32_t flags, void* payload, int64_t payload_stride, double* absmax,
                     int32_t* overflow_flag, void* stream, int fpc, long long clip_stride) {
    int rc = check_common(pcm, payload, n_frames, N, C, bits);
    if (rc == FRAD_OK && overflow_flag != nullptr && absmax == nullptr && n_frames > 0) return FRAD_E_INVALID;      // the test reads the per-frame maxima
    if (rc != FRAD_OK) return rc;
    if (!valid_dtype(pcm_dtype)) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (payload_stride < (int64_t)frad_payload_bytes(N, C, bits)) return FRAD_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lg = (pcm_dtype >> 1) & 3;
    const bool f32 = (pcm_dtype >> 3) == 2 && lg <= 2;
    Geom g = make_geom(n_frames, N, C, frame_stride, payload_stride, bits, flags, pcm_dtype);
    g.fpc = fpc; g.clip_stride = clip_stride;
    const int ai = (input_aligned(pcm, frame_stride, N, C, lg) && (fpc == 0 || ((clip_stride * C) << lg) % 16 == 0)) ? 1 : 0;
    const int ao = (aligned16(payload) && payload_stride % 16 == 0) ? 1 : 0;
    const unsigned char* in = static_cast<const unsigned char*>(pcm);
    unsigned char* out = static_cast<unsigned char*>(payload);
    {
        Geom gw = g;                                          // the wave kernels apply the overflow test themselves
        gw.ovf_flag = overflow_flag; gw.ovf_limit = storage_float_max(bits);
        if (launch_p0_fwd_wave(lg, s, in, out, absmax, gw, ai, ao, unit_neg)) { HIPCHK(hipGetLastError()); return FRAD_OK; }
    }
    if (fpc > 0 && (f32 || fast_cfg(N, C, f32).ok)) {
        // clips through a kernel without clip addressing: the clips' frame regions are gathered into a dense scratch first
        // (one strided device copy), then the flat batch runs on it
        const size_t row = ((size_t)fpc * N * C) << lg;
        StreamScratch ws{s};
        if (hipMallocAsync(&ws.p, row * (size_t)(n_frames / fpc), s) != hipSuccess) return FRAD_E_NOMEM;
        HIPCHK(hipMemcpy2DAsync(ws.p, row, pcm, ((size_t)clip_stride * C) << lg, row, (size_t)(n_frames / fpc), hipMemcpyDeviceToDevice, s));
        return p0_analogue_impl(ws.p, pcm_dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, overflow_flag, stream, 0, 0);
    }
    // every other kernel: the batch form of the test as a second launch on the same stream
    const int rc_all = [&]() -> int {
    const FastCfg c = fast_cfg(N, C, f32);
    if (c.ok) {
        Tables tb; rc = get_tables(c.log2m, f32, tb);
        if (rc != FRAD_OK) return rc;
        if (c.cg < C && bits == 12 && ((C & 1) || (c.cg & 1))) {       // pairs of values would straddle channel groups
            const int r = global_p0_analogue(in, out, absmax, g, flags, s);
            if (r == FRAD_E_HIP) g_last_hip = global_last_hip_error();
            return r;
        }
        g.fpb = c.fpb; g.cg = c.cg;
        if (c.cg == C && ai) {                               // quad stage-in needs whole 16-byte rows / row groups
            const int rb = C << lg;
            g.in_mode = (rb <= 4 && 16 % rb == 0) ? 1 : rb == 8 ? 2 : rb % 16 == 0 ? 3 : 0;
        }
        if (c.cg == C && ao && C <= 2 && ((long long)N * C) % 32 == 0) g.cc_fast = C;
        dim3 grid((unsigned)((n_frames + c.fpb - 1) / c.fpb));
        bool taken = false;
        if (f32 && c.log2m == 11 && C == 8 && c.cg == C) {      // BASELINE config 4's geometry: two half-frame blocks per frame
            if (absmax) HIPCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)n_frames, s));   // atomicMax target
            taken = launch_p0_fwd_half32(lg, c, s, in, out, absmax, tb, g, ai, ao) != 0;
        }
        if (taken) {
        } else if (!launch_p0_fwd_pers(f32, lg, c, s, in, out, absmax, tb, g, ao)) {
            if (absmax) HIPCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)n_frames, s));   // atomicMax target
            if (!f32 && c.cg < C && launch_p0_fwd_grp2(lg, c, s, in, out, absmax, tb, g, ai, ao)) {
                // two channel groups with whole-row I/O took it (frad_p0_fwd_grp2.hip)
            } else {
                rc = f32 ? launch_p0_fwd_f32(lg, c, grid, s, in, out, absmax, tb, g, ai, ao)
                         : launch_p0_fwd_f64(lg, c, grid, s, in, out, absmax, tb, g, ai, ao);
                if (rc != FRAD_OK) return rc;
            }
        }
    } else {
        {                                                    // N = 2 r 2^p, r in {3, 5, 7}: mixed-radix FFT (frad_mixed.hip), float64 for every PCM type
            const int r = launch_p0_fwd_mixed(lg, s, in, out, absmax, g, ai, ao, unit_neg);
            if (r < 0) { if (r == FRAD_E_HIP) g_last_hip = mixed_last_hip_error(); return r; }
            if (r == 1) return FRAD_OK;
        }
        if (!f32) {                                          // any N in O(N log N): Bluestein over the power-of-two FFT
            const int r = launch_p0_fwd_blue(lg, s, in, out, absmax, g, ao);
            if (r < 0) { if (r == FRAD_E_HIP) g_last_hip = blue_last_hip_error(); return r; }
            if (r == 1) { HIPCHK(hipGetLastError()); return FRAD_OK; }
        }
        if (fpc > 0) {                                       // (Bluestein did not take it) gather the clips, then the flat batch
            const size_t row = ((size_t)fpc * N * C) << lg;
            StreamScratch ws{s};
            if (hipMallocAsync(&ws.p, row * (size_t)(n_frames / fpc), s) != hipSuccess) return FRAD_E_NOMEM;
            HIPCHK(hipMemcpy2DAsync(ws.p, row, pcm, ((size_t)clip_stride * C) << lg, row, (size_t)(n_frames / fpc), hipMemcpyDeviceToDevice, s));
            return p0_analogue_impl(ws.p, pcm_dtype, n_frames, N, C, frame_stride, bits, flags, payload, payload_stride, absmax, nullptr, stream, 0, 0);
        }
        const size_t per_frame = 2 * (size_t)N * C * (f32 ? 4 : 8);
        if (per_frame > (size_t)kLdsBytes) {                 // wider than a CU's LDS: HBM workspaces (frad_global.hip)
            const int r = global_p0_analogue(in, out, absmax, g, flags, s);
            if (r == FRAD_E_HIP) g_last_hip = global_last_hip_error();
            return r;
        }
        DirectTable d; rc = get_direct(N, d);
        if (rc != FRAD_OK) return rc;
        if (absmax) HIPCHK(hipMemsetAsync(absmax, 0, sizeof(double) * (size_t)n_frames, s));
        g.fpb = direct_fpb(n_frames, C, per_frame);
        const size_t lds = per_frame * (size_t)g.fpb;
        const long long nblk = (n_frames + g.fpb - 1) / g.fpb;
        if (nblk > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
        dim3 grid((unsigned)nblk);
        const dim3 blk((unsigned)direct_threads(N));
#define FRAD_DIR(TT, LGV) do { allow_lds(k_p0_fwd_direct<TT, LGV>, lds); \
        hipLaunchKernelGGL((k_p0_fwd_direct<TT, LGV>), grid, blk, lds, s, in, out, absmax, d.ct, g, ai, ao); } while (0)
        if (f32) { if (lg == 1) FRAD_DIR(float, 1); else FRAD_DIR(float, 2); }
        else switch (lg) { case 0: FRAD_DIR(double, 0); break; case 1: FRAD_DIR(double, 1); break; case 2: FRAD_DIR(double, 2); break; default: FRAD_DIR(double, 3); break; }
#undef FRAD_DIR
    }
    HIPCHK(hipGetLastError());
    return FRAD_OK;
    }();
    if (rc_all != FRAD_OK) return rc_all;
    if (overflow_flag != nullptr) return frad_p0_overflow_scan(absmax, n_frames, bits, overflow_flag, stream);
    return FRAD_OK;
}

int p0_digital_impl(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                    uint32_t flags, double* pcm_out, void* stream, int fpc, long long clip_stride, int out_dtype = FRAD_PCM_F64LE);
}  // namespace
namespace frad {
// frad_p0_digital with the decoder's output conversion applied by the kernel's own store (frad_epilogue.hip: frad_p0_digital_pcm).
// FRAD_OK = done; 1 = this geometry's kernel cannot convert (the caller decodes to float64 scratch and converts in a second pass)
int p0_digital_out(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits, uint32_t flags,
                   int out_dtype, void* pcm_out, void* stream) {
    return p0_digital_impl(payload, payload_stride, n_frames, N, C, bits, flags, static_cast<double*>(pcm_out), stream, 0, 0, out_dtype);
}
}  // namespace frad
extern "C" {

int frad_p0_digital(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                    uint32_t flags, double* pcm_out, void* stream) {
    return p0_digital_impl(payload, payload_stride, n_frames, N, C, bits, flags, pcm_out, stream, 0, 0);
}

int frad_p0_digital_clips(const void* payload, int64_t payload_stride, int64_t n_clips, int32_t frames_per_clip, int32_t N, int32_t C,
                          int32_t bits, uint32_t flags, double* pcm_out, int64_t out_clip_stride, void* stream) {
    if (n_clips < 0 || frames_per_clip < 1 || N < 1 || out_clip_stride < (int64_t)frames_per_clip * N) return FRAD_E_INVALID;
    if (n_clips > 0x7fffffffLL / frames_per_clip) return FRAD_E_UNSUPPORTED;
    return p0_digital_impl(payload, payload_stride, n_clips * frames_per_clip, N, C, bits, flags, pcm_out, stream, frames_per_clip, out_clip_stride);
}

}  // extern "C"
namespace {

int p0_digital_impl(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                    uint32_t flags, double* pcm_out, void* stream, int fpc, long long clip_stride, int out_dtype) {
    const bool conv = out_dtype != FRAD_PCM_F64LE;            // the store converts (one-shot / channel-group / mixed-radix / Bluestein / direct kernels)
    if (conv && fpc > 0) return 1;
    int rc = check_common(payload, pcm_out, n_frames, N, C, bits);
    if (rc != FRAD_OK) return rc;
    if (n_frames == 0) return FRAD_OK;
    if (payload_stride < (int64_t)frad_payload_bytes(N, C, bits)) return FRAD_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Geom g = make_geom(n_frames, N, C, N, payload_stride, bits, flags, FRAD_PCM_F64LE);
    g.fpc = fpc; g.clip_stride = clip_stride;
    if (conv) { g.dtype = out_dtype; g.raw_be = (flags & FRAD_RAW_BE_INTS) ? 1 : 0; }
    const int ai = (aligned16(payload) && payload_stride % 16 == 0) ? 1 : 0;
    const unsigned char* in = static_cast<const unsigned char*>(payload);
    const int aout = (aligned16(pcm_out) && (fpc == 0 || ((clip_stride * C) * 8) % 16 == 0)) ? 1 : 0;
    if (!conv && launch_p0_inv_wave(s, in, pcm_out, g, ai, aout, unit_neg)) { HIPCHK(hipGetLastError()); return FRAD_OK; }
    const FastCfg c = fast_cfg(N, C, false);
    if (fpc > 0 && c.ok) {
        // clips through a kernel without clip addressing: the flat batch decodes into a dense scratch, one strided copy scatters it
        const size_t row = (size_t)fpc * N * C * 8;
        StreamScratch ws{s};
        if (hipMallocAsync(&ws.p, row * (size_t)(n_frames / fpc), s) != hipSuccess) return FRAD_E_NOMEM;
        rc = p0_digital_impl(payload, payload_stride, n_frames, N, C, bits, flags, static_cast<double*>(ws.p), stream, 0, 0);
        if (rc != FRAD_OK) return rc;
        HIPCHK(hipMemcpy2DAsync(pcm_out, (size_t)clip_stride * C * 8, ws.p, row, row, (size_t)(n_frames / fpc), hipMemcpyDeviceToDevice, s));
        return FRAD_OK;
    }
    if (c.ok) {
        Tables tb; rc = get_tables(c.log2m, false, tb);
        if (rc != FRAD_OK) return rc;
        // (12-bit pairs may straddle channel groups here: unpacking only reads them; the packing side refuses that)
        g.fpb = c.fpb; g.cg = c.cg;
        if (c.cg == C && ai && C <= 2 && ((long long)N * C) % 32 == 0) g.cc_fast = C;
        if (!conv && c.cg == C && C <= 2 && aligned16(pcm_out)) g.in_mode = C;      // decode: quad store for C = 1 / 2
        dim3 grid((unsigned)((n_frames + c.fpb - 1) / c.fpb));
        if (conv) {
            // every store converts, and the kernel is the one frad_p0_digital would run on an aligned float64 buffer (the result
            // must equal from_f64 of ITS samples bit for bit).  Only the wave kernel's geometries are left to the caller: its
            // converting twin (k_p0_inv_wave_pcm: s16 / s32 / f32) was tried first, the other formats take the second pass.
            if ((c.log2m == 9 || c.log2m == 10) && C <= 2) {
                if (p0_inv_wave_takes(g, ai, unit_neg)) return 1;
                if (c.cg == C) g.in_mode = C;                 // (what the float64 path sets for an aligned buffer: the unit kernels ask for it)
                if (launch_p0_inv_pers(c, s, in, pcm_out, tb, g)) { HIPCHK(hipGetLastError()); return FRAD_OK; }
                g.in_mode = 0;
            } else if (c.cg < C && launch_p0_inv_grp2(c, s, in, pcm_out, tb, g)) { HIPCHK(hipGetLastError()); return FRAD_OK; }
            rc = launch_p0_inv(c, grid, s, in, pcm_out, tb, g, ai);
            if (rc != FRAD_OK) return rc;
        } else if (c.cg < C && launch_p0_inv_grp2(c, s, in, pcm_out, tb, g)) {
            // whole-row two-pass kernel took it
        } else if (!launch_p0_inv_pers(c, s, in, pcm_out, tb, g)) {
            rc = launch_p0_inv(c, grid, s, in, pcm_out, tb, g, ai);
            if (rc != FRAD_OK) return rc;
        }
    } else {
        {
            const int rm = launch_p0_inv_mixed(s, in, pcm_out, g, ai, unit_neg);
            if (rm < 0) { if (rm == FRAD_E_HIP) g_last_hip = mixed_last_hip_error(); return rm; }
            if (rm == 1) return FRAD_OK;
        }
        const int r = launch_p0_inv_blue(s, in, pcm_out, g, ai);
        if (r < 0) { if (r == FRAD_E_HIP) g_last_hip = blue_last_hip_error(); return r; }
        if (r == 1) { HIPCHK(hipGetLastError()); return FRAD_OK; }
        if (fpc > 0) {                                       // (Bluestein did not take it) flat batch into a scratch, then scatter
            const size_t row = (size_t)fpc * N * C * 8;
            StreamScratch ws{s};
            if (hipMallocAsync(&ws.p, row * (size_t)(n_frames / fpc), s) != hipSuccess) return FRAD_E_NOMEM;
            rc = p0_digital_impl(payload, payload_stride, n_frames, N, C, bits, flags, static_cast<double*>(ws.p), stream, 0, 0);
            if (rc != FRAD_OK) return rc;
            HIPCHK(hipMemcpy2DAsync(pcm_out, (size_t)clip_stride * C * 8, ws.p, row, row, (size_t)(n_frames / fpc), hipMemcpyDeviceToDevice, s));
            return FRAD_OK;
        }
        const size_t per_frame = 2 * (size_t)N * C * 8;
        if (per_frame > (size_t)kLdsBytes) {
            if (conv) return 1;                              // (workspace path: float64 rows)
            const int r = global_p0_digital(in, pcm_out, g, flags, s);
            if (r == FRAD_E_HIP) g_last_hip = global_last_hip_error();
            return r;
        }
        DirectTable d; rc = get_direct(N, d);
        if (rc != FRAD_OK) return rc;
        g.fpb = direct_fpb(n_frames, C, per_frame);
        const size_t lds = per_frame * (size_t)g.fpb;
        const long long nblk = (n_frames + g.fpb - 1) / g.fpb;
        if (nblk > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
        allow_lds(k_p0_inv_direct<0>, lds);
        hipLaunchKernelGGL(k_p0_inv_direct<0>, dim3((unsigned)nblk), dim3((unsigned)direct_threads(N)), lds, s, in, pcm_out, d.ct, g, ai);
    }
    HIPCHK(hipGetLastError());
    return FRAD_OK;
}

}  // namespace
