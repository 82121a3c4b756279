// frad_launch.hpp -- host-side launch interface between the C-ABI translation unit (frad_hip.hip)
// and the translation units that instantiate the FFT kernels (frad_p0_*.hip).  The kernels are
// split over several objects only so that hipcc can build them in parallel.
#pragma once
#include "frad_kernels.hpp"
#include <vector>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace frad {

// FRAD_TUNE_* experiment knobs: the environment is read ONCE per call site and process (a function-local
// static per distinct use: no lock, no map on the launch path)
inline const char* tune_env(const char* name) { const char* e = std::getenv(name); return e ? (new std::string(e))->c_str() : nullptr; }
#define tune(name) ([]() -> const char* { static const char* const v_ = ::frad::tune_env(name); return v_; }())

struct Tables { void* tw = nullptr; void* post = nullptr; void* blob = nullptr; void* blob_b = nullptr; void* blob_i = nullptr; };   // blob: LDS image of the persistent kernels

// launch geometry of the LDS-resident FFT kernels
struct FastCfg {
    bool ok = false;
    int log2m = 0, team = 0, fpb = 0, threads = 0, cg = 0;
    size_t lds = 0;
};

struct DirectTable { double* ct = nullptr; };        // cos(pi j / 2N), j in [0, 4N)

// table caches and launch geometry (frad_hip.hip), shared with the profile-1 translation unit
int get_tables(int log2m, bool f32, Tables& out);
int get_direct(int N, DirectTable& out);
FastCfg fast_cfg(int N, int C, bool f32);

// opt a kernel in to more than 48 KiB of dynamic LDS -- once per kernel and size (the call is host-side
// work on every launch otherwise, and the GPU idles behind it between the encode and decode kernels)
template <typename K> inline void allow_lds(K kernel, size_t bytes) {
    static thread_local size_t granted = 0;          // one instance per kernel type K
    if (bytes > 48 * 1024 && bytes > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        granted = bytes;
    }
}

// each returns 0 or FRAD_E_UNSUPPORTED (-2) when that (log2m, lg) is not built
int launch_p0_fwd_f64_lo(int lg, const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                         double* absmax, const Tables& tb, const Geom& g, int aligned_in, int aligned_out);
int launch_p0_fwd_f64_hi(int lg, const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                         double* absmax, const Tables& tb, const Geom& g, int aligned_in, int aligned_out);
int launch_p0_fwd_f32(int lg, const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                      double* absmax, const Tables& tb, const Geom& g, int aligned_in, int aligned_out);
int launch_p0_inv(const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb,
                  const Geom& g, int aligned_in);

size_t pers_blob_build(int log2m, bool f32, int which, std::vector<unsigned char>& bytes,
                       void (*unit)(long long, long long, long double&, long double&));
// persistent kernels (frad_p0_pers.hip): return 1 when they took the launch, 0 when not applicable
int launch_p0_fwd_pers(bool f32, int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                       double* absmax, const Tables& tb, Geom g, int aligned_out);
int launch_p0_inv_pers(const FastCfg& c, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb, Geom g);
// two-pass whole-row encode / decode of frames with C = 2 * cg channels (frad_p0_fwd_grp2.hip, frad_p0_inv_grp2.hip): 1 = launched, 0 = not applicable
int launch_p0_fwd_grp2(int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax,
                       const Tables& tb, const Geom& g, int aligned_in, int aligned_out);
int launch_p0_inv_grp2(const FastCfg& c, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb, const Geom& g);
// float32 PCM, 8 channels at N = 4096: two co-resident half-frame blocks, one float32 pass each (frad_p0_fwd_half32.hip): 1 = launched
int launch_p0_fwd_half32(int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax,
                         const Tables& tb, const Geom& g, int aligned_in, int aligned_out);


// Bluestein kernels for frame lengths that are not a power of two (frad_p0_blue.hip): 1 = launched,
// 0 = not applicable (the caller falls back to the direct kernels), < 0 = FRAD_E_*
int launch_p0_fwd_blue(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax, Geom g, int aligned_out);
int launch_p0_inv_blue(hipStream_t s, const unsigned char* pay, double* out, Geom g, int aligned_in);
int blue_prepare(int N);
void blue_clear();
int blue_last_hip_error();
// mixed-radix kernels for N = 2 r 2^p, r in {3, 5, 7} (frad_mixed.hip): 1 = launched, 0 = not applicable, < 0 = FRAD_E_*
typedef void (*unit_root_fn)(long long, long long, long double&, long double&);      // exp(-i pi p / q)
int launch_p0_fwd_mixed(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax, const Geom& g,
                        int aligned_in, int aligned_out, unit_root_fn unit);
int launch_p0_inv_mixed(hipStream_t s, const unsigned char* pay, double* out, const Geom& g, int aligned_in, unit_root_fn unit);
int mixed_prepare(int N, unit_root_fn unit);
// the DCT of planar float64 rows in HBM through a complex workspace of rows * N / 2 slots (frames wider than a CU): 1 / 0 / < 0
int global_dct_mixed(bool fwd, const double* in, double* out, void* zw, int N, int C, long long rows, long long fstride, long long cstride,
                     long long ostride, hipStream_t s, unit_root_fn unit);
int p0_digital_out(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits, uint32_t flags,
                   int out_dtype, void* pcm_out, void* stream);     // frad_hip.hip: 0 = decoded and converted in one pass, 1 = not this geometry
int p1_digital_out(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C, int32_t bits, int32_t srate,
                   int out_dtype, uint32_t flags, void* out, void* stream);      // frad_p1.hip: as p0_digital_out
void unit_root(long long p, long long q, long double& re, long double& im);     // exp(-i pi p / q), q even (frad_hip.hip)
void mixed_clear();
int mixed_last_hip_error();
// wave-autonomous kernels for N = 2048, C <= 2, 16/32/64-bit storage (frad_p0_wave.hip): 1 = launched, 0 = not applicable
int launch_p0_fwd_wave(int lg, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* absmax, const Geom& g,
                       int aligned_in, int aligned_out, unit_root_fn unit);
bool p0_inv_wave_takes(const Geom& g, int aligned_in, unit_root_fn unit);
int launch_p0_inv_wave(hipStream_t s, const unsigned char* pay, double* out, const Geom& g, int aligned_in, int aligned_out,
                       unit_root_fn unit);
struct P1Wave;
int launch_p1_inv_wave(hipStream_t s, const int32_t* q, double* out, const Geom& g, const P1Wave& pw, unit_root_fn unit);
int launch_p1_fwd_wave(int lg, hipStream_t s, const unsigned char* pcm, int32_t* q, const Geom& g, const P1Wave& pw, int aligned_in, unit_root_fn unit);
int launch_p0_inv_wave_pcm(hipStream_t s, const unsigned char* pay, void* out, const Geom& g, int aligned_in, int out_dtype, unit_root_fn unit);
void wave_blob_build(std::vector<unsigned char>& bytes, unit_root_fn unit);
void wave_clear();
// workspace path of last resort (frad_global.hip): frames that no LDS-resident kernel can hold run through HBM buffers
int global_p0_analogue(const unsigned char* pcm, unsigned char* payload, double* absmax, const Geom& g, uint32_t flags, hipStream_t s);
int global_p0_digital(const unsigned char* payload, double* out, const Geom& g, uint32_t flags, hipStream_t s);
int global_last_hip_error();
// CRC-32 tables (frad_crc.hip)
void crc_clear();
void p1_clear();         // profile-1 band maps (frad_p1.hip)
int crc_last_hip_error();

}  // namespace frad
