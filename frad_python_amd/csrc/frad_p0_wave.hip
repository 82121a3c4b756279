// wave-autonomous profile-0 kernels (frad_wave.hpp): table blob, instantiation and launch policy
#include "frad_wave.hpp"
#include "frad_launch.hpp"
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace frad {
namespace {

int wave_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
// experiments only (tuning knobs): read once per process
bool wave_disabled() { static const bool d = [] { const char* e = tune("FRAD_TUNE_NO_WAVE"); return e && e[0] == '1'; }(); return d; }

// start stagger of the waves of a block, in steps of 64 cycles per (wave, block mod 8) slot (Geom::cg carries it; these kernels do not use cg)
int wave_stagger() { static const int st = [] { const char* e = tune("FRAD_TUNE_WAVE_STAGGER"); return e ? atoi(e) : 0; }(); return st; }

std::mutex g_wave_mu;
std::map<int, void*> g_wave_blob;                    // device -> LDS image of WaveLayout
int g_wave_hip = 0;

template <int LG, int CC, int BITS>
void go_fwd_wave(const void* blob, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am, const Geom& g) {
    if constexpr (LG == 1) {                                  // clip batches (frad_p0_analogue_clips): 16-bit PCM has the variant
        if (g.fpc > 0) {
            allow_lds(k_p0_fwd_wave<LG, CC, BITS, true>, kWaveLdsBytes);
            hipLaunchKernelGGL((k_p0_fwd_wave<LG, CC, BITS, true>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pcm, pay, am,
                               static_cast<const cx<double>*>(blob), g);
            return;
        }
    }
    allow_lds(k_p0_fwd_wave<LG, CC, BITS>, kWaveLdsBytes);
    hipLaunchKernelGGL((k_p0_fwd_wave<LG, CC, BITS>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pcm, pay, am,
                       static_cast<const cx<double>*>(blob), g);
}
template <int LG, int CC>
void go_fwd_wave_bits(const void* blob, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am, const Geom& g) {
    switch (g.bits) {
        case 16: go_fwd_wave<LG, CC, 16>(blob, grid, s, pcm, pay, am, g); break;
        case 32: go_fwd_wave<LG, CC, 32>(blob, grid, s, pcm, pay, am, g); break;
        default: go_fwd_wave<LG, CC, 64>(blob, grid, s, pcm, pay, am, g); break;
    }
}
template <int CC>
void go_fwd_wave_lg(int lg, const void* blob, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am, const Geom& g) {
#ifdef FRAD_WAVE_EXPERIMENT                                    // diagnostic builds: the headline element size only
    go_fwd_wave_bits<1, CC>(blob, grid, s, pcm, pay, am, g); (void)lg;
#else
    switch (lg) {
        case 0: go_fwd_wave_bits<0, CC>(blob, grid, s, pcm, pay, am, g); break;
        case 1: go_fwd_wave_bits<1, CC>(blob, grid, s, pcm, pay, am, g); break;
        case 2: go_fwd_wave_bits<2, CC>(blob, grid, s, pcm, pay, am, g); break;
        default: go_fwd_wave_bits<3, CC>(blob, grid, s, pcm, pay, am, g); break;
    }
#endif
}

template <int CC, int BITS>
void go_inv_wave(const void* blob, int grid, hipStream_t s, const unsigned char* pay, double* out, const Geom& g) {
    if (g.fpc > 0) {                                          // clip batches (frad_p0_digital_clips)
        allow_lds(k_p0_inv_wave<CC, BITS, true>, kWaveLdsBytes);
        hipLaunchKernelGGL((k_p0_inv_wave<CC, BITS, true>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pay, out,
                           static_cast<const cx<double>*>(blob), g);
        return;
    }
    allow_lds(k_p0_inv_wave<CC, BITS>, kWaveLdsBytes);
    hipLaunchKernelGGL((k_p0_inv_wave<CC, BITS>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pay, out,
                       static_cast<const cx<double>*>(blob), g);
}
template <int CC>
void go_inv_wave_bits(const void* blob, int grid, hipStream_t s, const unsigned char* pay, double* out, const Geom& g) {
    switch (g.bits) {
        case 16: go_inv_wave<CC, 16>(blob, grid, s, pay, out, g); break;
        case 32: go_inv_wave<CC, 32>(blob, grid, s, pay, out, g); break;
        default: go_inv_wave<CC, 64>(blob, grid, s, pay, out, g); break;
    }
}

}  // namespace

// Host image of WaveLayout.  `unit(p, q, re, im)` returns exp(-i pi p / q) (long double, exact octant symmetry).
void wave_blob_build(std::vector<unsigned char>& bytes, void (*unit)(long long, long long, long double&, long double&)) {
    constexpr int N = 2048;
    std::vector<cx<double>> out(WaveLayout::SLOTS);
    for (int k2 = 0; k2 < 32; ++k2)
        for (int l = 0; l < 32; ++l) {
            long double re, im; unit(2LL * l * k2, 1024, re, im);                 // W_1024^(l k2)
            out[WaveLayout::TW1 + k2 * 32 + l] = cx<double>{(double)re, (double)im};
        }
    auto wk = [&](int k) { long double re, im; unit(k, 2LL * N, re, im); return cx<double>{(double)re, (double)im}; };
    auto gk = [&](int k) { long double re, im; unit((long long)N + 5LL * k, 2LL * N, re, im); return cx<double>{(double)re, (double)im}; };
    out[WaveLayout::TW1 + 0] = wk(512);                                         // row 0 of TW1 is all ones and never read as such
    out[WaveLayout::TW1 + 1] = gk(512);
    for (int u = 0; u < 16; ++u)
        for (int a = 0; a < 32; ++a) {
            const int k = wave_job_k(a, u);
            out[WaveLayout::PW + u * 32 + a] = wk(k);
            out[WaveLayout::PG + u * 32 + a] = gk(k);
        }
    bytes.assign((unsigned char*)out.data(), (unsigned char*)(out.data() + out.size()));
}

static const void* wave_blob(void (*unit)(long long, long long, long double&, long double&)) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_wave_mu);
    auto it = g_wave_blob.find(dev);
    if (it != g_wave_blob.end()) return it->second;
    std::vector<unsigned char> bytes;
    wave_blob_build(bytes, unit);
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, bytes.size());
    if (e == hipSuccess) e = hipMemcpy(d, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { g_wave_hip = (int)e; return nullptr; }
    g_wave_blob[dev] = d;
    return d;
}
void wave_clear() {
    std::lock_guard<std::mutex> lk(g_wave_mu);
    for (auto& kv : g_wave_blob) (void)hipFree(kv.second);
    g_wave_blob.clear();
}

#if defined(FRAD_WAVE_STAMPS)
extern "C" int frad_debug_wave_stamps(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_wave_stamps), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

bool wave_geometry(int N, int C, int bits) { return N == 2048 && (C == 1 || C == 2) && (bits == 16 || bits == 32 || bits == 64); }

// 1 = launched, 0 = not this kernel's geometry (caller falls back to the unit / one-shot kernels)
int launch_p0_fwd_wave(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am, const Geom& g, int ai, int ao,
                       void (*unit)(long long, long long, long double&, long double&)) {
    if (wave_disabled() || !wave_geometry(g.N, g.C, g.bits) || !ai || !ao || g.n_valid != g.N) return 0;
    if (g.fpc > 0 && lg != 1) return 0;                       // clip batches: the 16-bit PCM variant only (others: strided copy + flat batch)
    if ((g.dtype >> 3) == 2 && ((g.dtype >> 1) & 3) <= 2) return 0;   // f16 / f32 PCM: float32 compute stays with the f32 kernels
    const void* blob = wave_blob(unit);
    if (blob == nullptr) return 0;
    const long long units = g.C == 2 ? g.n_frames : (g.n_frames + 1) / 2;
    const long long nb = (units + kWaveWaves - 1) / kWaveWaves, cap = wave_cu_count();
    const int grid = (int)(nb < cap ? nb : cap);
    Geom gg = g;
    gg.cg = wave_stagger();
    { static const int plain = [] { const char* e = tune("FRAD_TUNE_WAVE_PLAIN_PAYLOAD"); return e ? atoi(e) : 0; }(); gg.fpb = plain; }   // (fpb is unused by these kernels)
    if (g.C == 2) go_fwd_wave_lg<2>(lg, blob, grid, s, pcm, pay, am, gg);
    else go_fwd_wave_lg<1>(lg, blob, grid, s, pcm, pay, am, gg);
    return 1;
}

// would launch_p0_inv_wave take this batch into an aligned float64 buffer?  (frad_p0_digital_pcm must convert the samples of the
// kernel frad_p0_digital runs)
bool p0_inv_wave_takes(const Geom& g, int ai, unit_root_fn unit) {
    return !wave_disabled() && wave_geometry(g.N, g.C, g.bits) && ai && !tune("FRAD_TUNE_NO_WAVE_DEC") && wave_blob(unit) != nullptr;
}

int launch_p0_inv_wave(hipStream_t s, const unsigned char* pay, double* out, const Geom& g, int ai, int ao, unit_root_fn unit) {
    if (wave_disabled() || !wave_geometry(g.N, g.C, g.bits) || !ai || !ao) return 0;
    if (tune("FRAD_TUNE_NO_WAVE_DEC")) return 0;
    const void* blob = wave_blob(unit);
    if (blob == nullptr) return 0;
    const long long units = g.C == 2 ? g.n_frames : (g.n_frames + 1) / 2;
    const long long nb = (units + kWaveWaves - 1) / kWaveWaves, cap = wave_cu_count();
    const int grid = (int)(nb < cap ? nb : cap);
    Geom gg = g;
    gg.cg = wave_stagger();
    { static const int plain = [] { const char* e = tune("FRAD_TUNE_WAVE_PLAIN_LOADS"); return e ? atoi(e) : 0; }(); gg.fpb = plain; }
    if (g.C == 2) go_inv_wave_bits<2>(blob, grid, s, pay, out, gg);
    else go_inv_wave_bits<1>(blob, grid, s, pay, out, gg);
    return 1;
}


// shared with the profile-1 wave kernels (frad_p1_wave.hip)
const void* wave_blob_get(unit_root_fn unit) { return wave_blob(unit); }
int wave_grid(long long units) {
    const long long nb = (units + kWaveWaves - 1) / kWaveWaves, cap = wave_cu_count();
    return (int)(nb < cap ? nb : cap);
}
bool wave_off() { return wave_disabled(); }
int wave_stagger_steps() { return wave_stagger(); }

}  // namespace frad
