// profile-0 wave decode with the from_f64 output conversion fused (frad_wave.hpp, OUT >= 0): N = 2048 stereo or mono, 16- / 32-bit
// storage, s16le / s32le / f32le out.  1 = launched, 0 = not applicable (frad_p0_digital_pcm then decodes to float64 scratch
// and narrows in a second pass).
#include "frad_wave.hpp"
#include "frad_launch.hpp"
#include "../../include/frad_hip.h"

namespace frad {

const void* wave_blob_get(unit_root_fn unit);
int wave_grid(long long units);
bool wave_off();
int wave_stagger_steps();

namespace {
template <int BITS, int OUT>
void go(const void* blob, int grid, hipStream_t s, const unsigned char* pay, void* out, const Geom& g) {
    if (g.C == 1) {
        allow_lds(k_p0_inv_wave_pcm<BITS, OUT, 1>, kWaveLdsBytes);
        hipLaunchKernelGGL((k_p0_inv_wave_pcm<BITS, OUT, 1>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pay, out, static_cast<const cx<double>*>(blob), g);
        return;
    }
    allow_lds(k_p0_inv_wave_pcm<BITS, OUT>, kWaveLdsBytes);
    hipLaunchKernelGGL((k_p0_inv_wave_pcm<BITS, OUT>), dim3(grid), dim3(64 * kWaveWaves), kWaveLdsBytes, s, pay, out, static_cast<const cx<double>*>(blob), g);
}
template <int OUT>
bool go_bits(const void* blob, int grid, hipStream_t s, const unsigned char* pay, void* out, const Geom& g) {
    if (g.bits == 16) go<16, OUT>(blob, grid, s, pay, out, g);
    else if (g.bits == 32) go<32, OUT>(blob, grid, s, pay, out, g);
    else return false;
    return true;
}
}  // namespace

int launch_p0_inv_wave_pcm(hipStream_t s, const unsigned char* pay, void* out, const Geom& g, int ai, int out_dtype, unit_root_fn unit) {
    static const bool off = [] { const char* e = tune("FRAD_TUNE_NO_WAVE_PCM"); return e && e[0] == '1'; }();
    if (wave_off() || off || g.N != 2048 || (g.C != 2 && g.C != 1) || !ai || (reinterpret_cast<uintptr_t>(out) & 15)) return 0;
    if (out_dtype != FRAD_PCM_S16LE && out_dtype != FRAD_PCM_S32LE && out_dtype != FRAD_PCM_F32LE) return 0;
    if (g.bits != 16 && g.bits != 32) return 0;
    const void* blob = wave_blob_get(unit);
    if (blob == nullptr) return 0;
    Geom gg = g;
    gg.cg = wave_stagger_steps(); gg.fpb = 0;
    const int grid = wave_grid(g.C == 2 ? g.n_frames : (g.n_frames + 1) / 2);    // unit = a stereo frame or two mono frames
    bool ok = false;
    if (out_dtype == FRAD_PCM_S16LE) ok = go_bits<FRAD_PCM_S16LE>(blob, grid, s, pay, out, gg);
    else if (out_dtype == FRAD_PCM_S32LE) ok = go_bits<FRAD_PCM_S32LE>(blob, grid, s, pay, out, gg);
    else ok = go_bits<FRAD_PCM_F32LE>(blob, grid, s, pay, out, gg);
    return ok ? 1 : 0;
}

}  // namespace frad
