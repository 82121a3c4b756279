// profile 1 on the wave-autonomous kernels (frad_wave.hpp, MODE 1): K7 / K8 at N = 2048, one or two channels.
// Instantiation and launch policy; 1 = launched, 0 = not applicable (the caller falls back to the one-shot kernels).
#include "frad_wave.hpp"
#include "frad_launch.hpp"

namespace frad {

const void* wave_blob_get(unit_root_fn unit);
int wave_grid(long long units);
bool wave_off();
int wave_stagger_steps();

namespace {
bool p1_wave_off() { static const bool d = [] { const char* e = tune("FRAD_TUNE_NO_WAVE_P1"); return e && e[0] == '1'; }(); return d; }

template <int LG, int CC>
void go_p1_fwd(const void* blob, int grid, hipStream_t s, const unsigned char* pcm, int32_t* q, const Geom& g, const P1Wave& pw) {
    allow_lds(k_p1_fwd_wave<LG, CC>, kWaveLdsBytesP1);
    hipLaunchKernelGGL((k_p1_fwd_wave<LG, CC>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytesP1, s, pcm, q, static_cast<const cx<double>*>(blob), g, pw);
}
template <int CC>
void go_p1_fwd_lg(int lg, const void* blob, int grid, hipStream_t s, const unsigned char* pcm, int32_t* q, const Geom& g, const P1Wave& pw) {
    switch (lg) {
        case 1: go_p1_fwd<1, CC>(blob, grid, s, pcm, q, g, pw); break;
        case 2: go_p1_fwd<2, CC>(blob, grid, s, pcm, q, g, pw); break;
        default: go_p1_fwd<3, CC>(blob, grid, s, pcm, q, g, pw); break;
    }
}
}  // namespace

#if defined(FRAD_WAVE_STAMPS)
extern "C" int frad_debug_wave_stamps_p1(unsigned long long* out, int reset) {      // this translation unit's copy of the counters
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_wave_stamps), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

// K7.  Needs whole frames (n_valid = N), 16-byte aligned PCM rows (ai), integer or float64 PCM of 2, 4 or 8 bytes.
int launch_p1_fwd_wave(int lg, hipStream_t s, const unsigned char* pcm, int32_t* q, const Geom& g, const P1Wave& pw, int ai, unit_root_fn unit) {
    if (wave_off() || p1_wave_off() || g.N != 2048 || (g.C != 1 && g.C != 2) || !ai || g.n_valid != g.N || lg < 1) return 0;
    if ((g.dtype >> 3) == 2 && lg <= 2) return 0;              // f16 / f32 PCM: the reference's mixed-precision path (one-shot kernels)
    if (pw.edge[26] < g.N || (reinterpret_cast<uintptr_t>(q) & 15) || pw.tqh == nullptr) return 0;
    // the tail's band-energy pass (wave_p1_tail): a run of 32 consecutive bins touches at most three bands and a band spans at
    // most kK7Slots runs -- true for every table rate up to 48 kHz at this frame length
    for (int b = 0; b < 26 && pw.edge[b] < g.N; ++b) {
        if (pw.edge[b + 1] <= pw.edge[b]) return 0;                       // (bands are non-empty up to the one that holds bin N - 1)
        const int last = (pw.edge[b + 1] < g.N ? pw.edge[b + 1] : g.N) - 1;
        if ((last >> 5) - (pw.edge[b] >> 5) + 1 > kK7Slots) return 0;
    }
    auto band_at = [&](int k) { int b = 0; while (b < 25 && pw.edge[b + 1] <= k) ++b; return b; };
    for (int r = 0; r < g.N / 32; ++r) if (band_at(32 * r + 31) - band_at(32 * r) > 2) return 0;
    const void* blob = wave_blob_get(unit);
    if (blob == nullptr) return 0;
    Geom gg = g;
    gg.cg = wave_stagger_steps(); gg.fpb = 0; gg.bits = 32; gg.le = 1; gg.payload_stride = (long long)g.N * g.C * 4;
    gg.ovf_flag = nullptr; gg.ovf_limit = 0.0;
    const int grid = wave_grid(g.C == 2 ? g.n_frames : (g.n_frames + 1) / 2);
    if (g.C == 2) go_p1_fwd_lg<2>(lg, blob, grid, s, pcm, q, gg, pw);
    else go_p1_fwd_lg<1>(lg, blob, grid, s, pcm, q, gg, pw);
    return 1;
}

// K8
int launch_p1_inv_wave(hipStream_t s, const int32_t* q, double* out, const Geom& g, const P1Wave& pw, unit_root_fn unit) {
    if (wave_off() || p1_wave_off() || g.N != 2048 || (g.C != 1 && g.C != 2)) return 0;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(q) & 3) || pw.edge[26] < g.N || pw.deq == nullptr) return 0;
    const void* blob = wave_blob_get(unit);
    if (blob == nullptr) return 0;
    const int grid = wave_grid(g.C == 2 ? g.n_frames : (g.n_frames + 1) / 2);
    Geom gg = g;
    gg.cg = wave_stagger_steps(); gg.fpb = 0; gg.bits = 32; gg.le = 1; gg.payload_stride = (long long)g.N * g.C * 4;
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(q);
    if (g.C == 2) {
        allow_lds(k_p1_inv_wave<2>, kWaveLdsBytesP1);
        hipLaunchKernelGGL(k_p1_inv_wave<2>, dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytesP1, s, qb, out, static_cast<const cx<double>*>(blob), gg, pw);
    } else {
        allow_lds(k_p1_inv_wave<1>, kWaveLdsBytesP1);
        hipLaunchKernelGGL(k_p1_inv_wave<1>, dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytesP1, s, qb, out, static_cast<const cx<double>*>(blob), gg, pw);
    }
    return 1;
}

}  // namespace frad
