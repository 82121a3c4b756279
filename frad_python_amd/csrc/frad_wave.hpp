// frad_wave.hpp -- wave-autonomous profile-0 kernels for N = 2048, float64 compute, 1 or 2 channels.
//
// Reference path: /root/reference/src/libfrad/fourier/profile0.py:14-44 (analogue) and :46-69 (digital); the
// DCT is scipy.fft.dct/idct(norm='forward') at profile0.py:21,69; to_f64 is backend/pcmformat.py:34-47.
//
// Why another kernel family.  The unit kernels (frad_persistent.hpp) give one wave one channel-frame with 16
// points per lane: three radix passes, i.e. FOUR trips of the whole 16 KiB channel buffer through LDS (stage-in,
// two exchanges, pack-out), an inter-wave barrier three times per frame, and 158 KiB of LDS for 8 waves.  rocprofv3
// (round 1) showed them issue- and LDS-write-bound at 0.31 of the HBM roofline.  Here ONE WAVE OWNS ONE STEREO
// FRAME (or two mono frames): lanes 0-31 hold channel 0, lanes 32-63 channel 1, 32 complex points per lane, so
// that M = 1024 = 32 x 32 needs a single exchange:
//
//   stage raw PCM bytes in LDS (16-byte copies, no conversion)      -> each lane picks its Makhoul pairs
//   pass 1: 32-point DFT in registers over z[l + 32 j], twiddle W_1024^(l k2)
//   ONE exchange through LDS (XOR-swizzled, conflict-free both ways), real plane then imaginary plane, so that a
//   wave needs 16 KiB instead of 32 and eight waves (two per SIMD) share a CU
//   pass 2: lane a needs Z[a + 32 m] AND its DCT partners Z[M - k], which live in residue -a.  It therefore
//           computes the even-m half of residue a and the odd-m half of residue -a (a 32-point DFT splits into
//           two 16-point DFTs of the sums and the twiddled differences: half the outputs for half the work), so
//           both members of every (k, M-k) pair meet in one lane and the DCT pair step runs on registers.  The
//           price is reading the exchange buffer twice (LDS reads are 3x cheaper than writes on CDNA4).
//   pack: storage codes go to LDS at their payload position, come back as 16-byte rows, coalesced stores.
//
// No s_barrier, no inter-wave traffic: the LDS serves one wave's instructions in order, so phases only need the
// compiler kept honest (team_sync<64>).  Two waves per SIMD (8 x 16 KiB + 32 KiB of tables = 160 KiB), 256 VGPRs
// each; the next frame's PCM is prefetched into registers during the transform.
#pragma once
#include "frad_kernels.hpp"
#include <type_traits>

namespace frad {

// LDS table blob (complex<double> slots), shared by the eight waves of a block:
//   TW1[k2 * 32 + l] = W_1024^(l * k2)                (row 0 is never used as a twiddle: slots 0, 1 hold w_512, g_512)
//   PW [u * 32 + a]  = w_k, PG[u * 32 + a] = g_k     for the pair job of (lane a, slot u), k = wave_job_k(a, u)
// The kernels scale w_k, g_k by the exact power of two 1/2N (x the deferred PCM normalisation) while copying them in.
struct WaveLayout { static constexpr int TW1 = 0, PW = 1024, PG = 1536, SLOTS = 2048; };
constexpr int kWaveTableBytes = WaveLayout::SLOTS * 16;
constexpr int kWavePlaneSlots = 34 * 32;                 // one padded exchange plane: 32 rows of 32 doubles + 2 (see pslot)
constexpr int kWaveBufBytes = 2 * kWavePlaneSlots * 8;   // 17 408 B: two planes (one per channel); the raw bytes and the payload staging alias them
constexpr int kWaveWaves = 7;                            // 7 x 17 KiB + 32 KiB of tables = 151 of the 160 KiB (an eighth wave -- tables cut to
                                                         // 24 KiB -- was measured in round 3: encode 92.4 -> 90.6 us, decode 113.8 -> 116.6 us; not kept)
constexpr int kWaveLdsBytes = kWaveTableBytes + kWaveWaves * kWaveBufBytes + 16;   // + the block's work counter

// pair job (lane a in [0, 32), slot u in [0, 16)) -> its k in [0, M/2]; see the header comment and pass 2 below
__host__ __device__ constexpr int wave_job_k(int a, int u) {
    return u < 8 ? a + 64 * u : (a == 0 ? 992 - 64 * u : 1024 - a - 64 * u);
}

template <typename T> struct W32K {
    // cos(pi e / 16), e = 0 .. 8, correctly rounded
    static constexpr T c[9] = {(T)1.0L, (T)0.98078528040323044912618223613424L, (T)0.92387953251128675612818318939679L,
                               (T)0.83146961230254523707878837761791L, (T)0.70710678118654752440084436210485L,
                               (T)0.55557023301960222474283081394853L, (T)0.38268343236508977172845998403040L,
                               (T)0.19509032201612826784828486847702L, (T)0.0L};
};
// a * W_32^E (forward) or its conjugate (inverse), E in [0, 16)
template <int E, bool INV, typename T> __device__ __forceinline__ cx<T> mul_w32(cx<T> a) {
    static_assert(E >= 0 && E < 16, "twiddle exponent");
    if constexpr (E == 0) return a;
    else if constexpr (E == 8) return mul_mi<INV>(a);
    else if constexpr (E == 4) return mul_w16<2, INV>(a);
    else if constexpr (E == 12) return mul_w16<6, INV>(a);
    else {
        constexpr T c = E < 8 ? W32K<T>::c[E] : -W32K<T>::c[16 - E];
        constexpr T s = W32K<T>::c[E < 8 ? 8 - E : E - 8];
        const cx<T> w = {c, INV ? s : -s};
        return cmul(a, w);
    }
}
// decimation in frequency, first stage of a 32-point DFT: e[n] = lo[n] + hi[n], o[n] = (lo[n] - hi[n]) W_32^n; the
// 16-point DFTs of e and o are the even and the odd outputs
template <bool INV, typename T, int I = 0>
__device__ __forceinline__ void dif32_stage(const cx<T> (&z)[32], cx<T> (&e)[16], cx<T> (&o)[16]) {
    if constexpr (I < 16) {
        e[I] = z[I] + z[I + 16];
        o[I] = mul_w32<I, INV>(z[I] - z[I + 16]);
        dif32_stage<INV, T, I + 1>(z, e, o);
    }
}
template <bool INV, typename T, int I = 0>
__device__ __forceinline__ void tw32_apply(cx<T> (&o)[16]) {          // o[n] *= W_32^n
    if constexpr (I < 16) { o[I] = mul_w32<I, INV>(o[I]); tw32_apply<INV, T, I + 1>(o); }
}

// Exchange plane of 32 x 32 doubles, rows padded to 34: (row r, column c) lives at 34 r + 2 (c & 15) + (c >> 4), i.e.
// columns c and c + 16 are neighbours.  Writers hold c = lane and sweep r (ds_write_b64: 16 consecutive lanes cover 64
// banks), readers hold r = lane and fetch the pair (c, c + 16) -- exactly what the first stage of pass 2 adds and
// subtracts -- with one ds_read_b128 (row pitch 272 B = 4 banks mod 64: 16 lanes cover 64 banks).  No conflicts
// either way, and -- unlike an XOR swizzle -- the swept index stays in the instruction's immediate offset: three
// address registers per wave instead of ~100.
__device__ __forceinline__ int pslot(int r, int c) { return 34 * r + 2 * (c & 15) + (c >> 4); }

// Diagnostic build only (-DFRAD_WAVE_STAMPS): per-phase shader-clock totals of wave 0 of every block, summed into
// g_wave_stamps[phase] (cycles, phases 0 .. 11) and g_wave_stamps[15] (units); no stamp executes in the product build.
#if defined(FRAD_WAVE_STAMPS) && !defined(FRAD_HOST_EMULATION)
__device__ unsigned long long g_wave_stamps[16];
#define FRAD_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#define FRAD_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime(), st_units = 0
#define FRAD_STAMP_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_wave_stamps[i_], st_acc[i_]); atomicAdd(&g_wave_stamps[15], st_units); } } while (0)
#else
#define FRAD_STAMP(i) ((void)0)
#define FRAD_STAMP_DECL ((void)0)
#define FRAD_STAMP_FLUSH ((void)0)
#endif

// section marks for tools/asm_stats.py: a comment in the assembly, no instruction -- but an asm statement the scheduler will not
// move code across, so diagnostic builds only (hipcc -DFRAD_ASM_MARKS --cuda-device-only -S ...)
#if defined(FRAD_ASM_MARKS) && !defined(FRAD_HOST_EMULATION)
#define FRAD_MARK(name) asm volatile("; FRAD_MARK " name)
#else
#define FRAD_MARK(name) ((void)0)
#endif
// work distribution inside a block: one LDS word, bumped by lane 0 of a wave and broadcast
#ifndef FRAD_HOST_EMULATION
#define FRAD_WAVE_LDS_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FRAD_WAVE_BCAST0(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
#define FRAD_WAVE_SLEEP(n) __builtin_amdgcn_s_sleep(n)
#else
#define FRAD_WAVE_LDS_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define FRAD_WAVE_BCAST0(x) ((unsigned)__shfl((unsigned long long)(x), 0, 64))
#define FRAD_WAVE_SLEEP(n) ((void)0)
#endif
__device__ __forceinline__ long long wave_next_unit(unsigned* ctr) {
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = FRAD_WAVE_LDS_ADD(ctr, 1u);
    return (long long)FRAD_WAVE_BCAST0(v);
}

#ifndef FRAD_HOST_EMULATION
#define FRAD_WAVE_BOUNDS __launch_bounds__(64 * kWaveWaves, 2)
#else
#define FRAD_WAVE_BOUNDS
#endif

// maxima of the two 32-lane halves of a wave (all lanes get both)
__device__ __forceinline__ void half_wave_max_u64(u64 v, u64& lo, u64& hi) {
    auto op = [](u64 a, u64 b) { return b > a ? b : a; };
#ifdef FRAD_HOST_EMULATION
    for (int off = 1; off < 32; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    const u64 other = __shfl_xor(v, 32, 64);
    const bool up = (threadIdx.x & 32) != 0;
    lo = up ? other : v; hi = up ? v : other;
#else
    v = op(v, dpp_move_u64<0xB1>(v));
    v = op(v, dpp_move_u64<0x4E>(v));
    v = op(v, dpp_move_u64<0x141>(v));
    v = op(v, dpp_move_u64<0x140>(v));
    lo = op(read_lane_u64(v, 0), read_lane_u64(v, 16));
    hi = op(read_lane_u64(v, 32), read_lane_u64(v, 48));
#endif
}

// running max of |v| in one instruction (fmax(fm, fabs(v)) costs two: the compiler canonicalises the operand first)
__device__ __forceinline__ void absmax_acc(double& fm, double v) {
#ifdef FRAD_HOST_EMULATION
    fm = fmax(fm, fabs(v));
#else
    asm("v_max_f64 %0, %1, |%2|" : "=v"(fm) : "v"(fm), "v"(v));
#endif
}

// 16-byte store of data that is written once and not read again by this kernel (payload rows, decoded PCM): nontemporal
__device__ __forceinline__ void stream_store(unsigned char* p, v4u v) {
#if defined(FRAD_HOST_EMULATION) || defined(FRAD_WAVE_PLAIN_STORES)
    *FRAD_GPTR(v4u, p) = v;
#else
    __builtin_nontemporal_store(v, FRAD_GPTR(v4u, p));
#endif
}

// bytes of `v` picked by a v_perm_b32 selector (selector bytes 4..7 address v's bytes 0..3)
__device__ __forceinline__ uint32_t wave_perm(uint32_t v, uint32_t sel) {
#ifdef FRAD_HOST_EMULATION
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) { const uint32_t s = (sel >> (8 * i)) & 0xffu; r |= (s >= 4 && s < 8 ? (v >> (8 * (s - 4))) & 0xffu : 0u) << (8 * i); }
    return r;
#else
    return __builtin_amdgcn_perm(v, 0u, sel);
#endif
}

// EB-bit element at bit `shift` of `word` -> compute type, format CODE (cvt_pcm_c semantics, normalisation deferred)
template <int CODE, bool RAW, int EB>
__device__ __forceinline__ double wave_elem(uint32_t word, int shift) {
    constexpr int kind = CODE >> 3, be = CODE & 1;
    if constexpr (EB == 32) {
        return cvt_pcm_c<double, CODE, RAW, true>((u64)word);
    } else if constexpr (kind == 1 && !be) {                 // signed little-endian: one v_bfe_i32 + one conversion
#ifdef FRAD_HOST_EMULATION
        const int s = (int)((word >> shift) << (32 - EB)) >> (32 - EB);
#else
        const int s = __builtin_amdgcn_sbfe((int)word, (unsigned)shift, (unsigned)EB);
#endif
        return (double)s;
    } else {
#ifdef FRAD_HOST_EMULATION
        const uint32_t r = (word >> shift) & ((1u << EB) - 1u);
#else
        const uint32_t r = __builtin_amdgcn_ubfe(word, (unsigned)shift, (unsigned)EB);
#endif
        return cvt_pcm_c<double, CODE, RAW, true>((u64)r);
    }
}

// ---- profile 1 on the same wave-per-frame structure (MODE 1 of the kernel bodies below) ------------------------------
// K8 = the decode body with the payload words replaced by the quantised integers (q has the layout of a 32-bit payload:
// bin-major, channel-minor, 4 bytes) and `value` = dequantise x threshold ramp (profile1.py:65-77); K7 = the encode body
// with the storage cast replaced by band energies -> thresholds -> per-bin divide + power-law quantiser (profile1.py:
// 15-40, p1tools.py:15-44).  Per-launch constants arrive by value and are copied to LDS once per block.
struct P1Wave {
    int edge[28];                // band edges in bins, clipped to N (p1tools.py:15-16)
    double floor_[27];           // min(ATH, 1.0) per band
    double scale, loss;          // 2^(bits-1); max(|loss_level|, 0.125)
    int nb_used;                 // bands before the first empty one
    const unsigned char* band_of;// device table [N]: band of bin k (0..25), 255 = beyond the last band start
    const double* deq;           // device table [512]: a^(1/0.75) for a = 0 .. 255 (p1tools.py:44), then (e/2)^(t^0.75) for
                                 // t = 0 .. 255 (profile1.py:63), both correctly rounded
    const int32_t* tq_in;        // K8: [n_frames, 27, C]
    int32_t* tq_out;             // K7: [n_frames, 27, C]
    const double* tqh;           // K7: device table [256]: (e/2)^((n + 1/2)^0.75), the thresholds at which the band code of
                                 // profile1.py:38-40 steps from n to n + 1 (correctly rounded)
    int* redo;                   // K8: redo[0] = number of frames the table-driven kernel could not take (a band code outside
                                 // [0, 256) or |q| >= 256: corrupt or extremely loud frames), redo[1 + i] = their indices; the
                                 // exact one-shot kernel decodes them again behind it (k_p1_inv_redo)
};
struct P1None {};
// after the work counter, K7: band_of[2048] | edge[32] | floor[32] | pk[2048] shorts | tqh[256] doubles; K8: see P1K8Lds
constexpr int kP1BlockBytes = 2048 + 32 * 4 + 32 * 8 + 2048 * 2 + 256 * 8;
constexpr int kWaveLdsBytesP1 = kWaveLdsBytes + 16 + kP1BlockBytes;
static_assert(kWaveLdsBytesP1 <= 160 * 1024, "a CU's LDS");
__device__ __forceinline__ void p1w_tables_to_lds(unsigned char* smem, const P1Wave& pw) {      // K7
    unsigned char* b = smem + kWaveLdsBytes + 16;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) reinterpret_cast<uint32_t*>(b)[i] = reinterpret_cast<const uint32_t*>(pw.band_of)[i];
    if (threadIdx.x < 28) reinterpret_cast<int*>(b + 2048)[threadIdx.x] = pw.edge[threadIdx.x];
    if (threadIdx.x < 27) reinterpret_cast<double*>(b + 2048 + 128)[threadIdx.x] = pw.floor_[threadIdx.x];
    if (pw.tqh != nullptr) for (int i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<double*>(b + 2048 + 128 + 256 + 4096)[i] = pw.tqh[i];
}
// K8's block tables (same area): edge[32] ints | pk[2048] shorts (as K7's, see p1w_k7_tables) | deqs[256] doubles =
// a^(1/0.75) / 2^(bits-1) (exact: the scale is a power of two) | thrt[256] doubles = (e/2)^(t^0.75)
struct P1K8Lds { const int* edge; const unsigned short* pk; const double* deqs; const double* thrt; };
constexpr int kK8EdgeOff = 0, kK8PkOff = 128, kK8DeqOff = 128 + 4096, kK8ThrOff = 128 + 4096 + 2048;
static_assert(kK8ThrOff + 2048 <= kP1BlockBytes, "K8 tables");
__device__ __forceinline__ P1K8Lds p1w_k8_lds(unsigned char* smem) {
    unsigned char* b = smem + kWaveLdsBytes + 16;
    return {reinterpret_cast<const int*>(b + kK8EdgeOff), reinterpret_cast<const unsigned short*>(b + kK8PkOff),
            reinterpret_cast<const double*>(b + kK8DeqOff), reinterpret_cast<const double*>(b + kK8ThrOff)};
}
__device__ __forceinline__ int k7_bin_of(int s, int cls, int l);
// phase 1 (before a block barrier): edges and the two value tables; phase 2 (after it): the bin -> (band, position) map
__device__ __forceinline__ void p1w_k8_tables_a(unsigned char* smem, const P1Wave& pw, double inv_scale) {
    unsigned char* b = smem + kWaveLdsBytes + 16;
    if (threadIdx.x < 28) reinterpret_cast<int*>(b + kK8EdgeOff)[threadIdx.x] = pw.edge[threadIdx.x];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        reinterpret_cast<double*>(b + kK8DeqOff)[i] = pw.deq[i] * inv_scale;
        reinterpret_cast<double*>(b + kK8ThrOff)[i] = pw.deq[256 + i];
    }
}
__device__ __forceinline__ void p1w_k8_tables_b(unsigned char* smem, const P1Wave& pw) {
    unsigned char* b = smem + kWaveLdsBytes + 16;
    const int* edge = reinterpret_cast<const int*>(b + kK8EdgeOff);
    unsigned short* pk = reinterpret_cast<unsigned short*>(b + kK8PkOff);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const int cls = i & 3, l = (i >> 2) & 31, s = i >> 7;
        const int kb = k7_bin_of(s, cls, l), j = pw.band_of[kb];
        pk[i] = (unsigned short)(j < 26 ? ((kb - edge[j]) | (j << 10)) : (31 << 10));      // band 31: {0, 0} (bins past the last band start)
    }
}
__device__ __forceinline__ double p1w_quant(double x) {        // sign(x) |x|^0.75 (p1tools.py:43)
    const double a = fabs(x), r = sqrt(a);
    return copysign(r * sqrt(r), x) * (a != 0.0);
}

// (int) round(sign(m) |m * scale|^0.75) with m = x / div -- the per-bin quantiser of profile1.py:27-36 -- decided in
// float32 whenever that is safe: the float32 value of |.|^0.75 carries a relative error below 1.5e-6 (two conversions,
// a reciprocal, two square roots, four products), so unless it lies within that distance of a half-integer it rounds to
// the same integer as the float64 value; the rare undecided bin (about y x 3e-6 of them) takes the exact path.  The
// result is therefore identical to the float64 arithmetic's, at a quarter of its instructions.
__device__ __forceinline__ float p1w_sqrtf(float v) {
#ifdef FRAD_HOST_EMULATION
    return sqrtf(v);
#else
    return __builtin_amdgcn_sqrtf(v);
#endif
}
__device__ __forceinline__ int32_t p1w_quantise(double x, double div, double scale) {
    const float xf = (float)x, df = (float)div;
    if (div == 0.0) return 0;                                 // x / inf -> +-0 -> 0
    const float mf = fabsf(xf) / df * (float)scale;           // div > 0: thresholds are non-negative
    if (mf < 1e30f) {                                         // (false for NaN: those take the exact path)
        const float r = p1w_sqrtf(mf), yf = r * p1w_sqrtf(r);
        const float fr = yf - floorf(yf);
        if (fabsf(fr - 0.5f) > yf * 3e-6f + 1e-6f) {
            const int32_t qa = (int32_t)rintf(yf);
            return xf < 0.0f ? -qa : qa;
        }
    }
    const double m = x / div;
    return (int32_t)rint(p1w_quant(m * scale));
}
// Band energy -> masking threshold of one band (p1tools.py:18-33).  `f32`: numpy's types on float32 coefficients --
// mean, sqrt and the 0.8 power stay float32, and the product with the loss level does too when the signal term wins.
__device__ __forceinline__ double p1_band_threshold(double energy, int bins, double floor_, double loss, int f32) {
    if (!f32) {
        const double sfq = pow(sqrt(energy / (double)bins), 0.8);
        return (floor_ > sfq ? floor_ : sfq) * loss;
    }
    const float mean = (float)(energy / (double)bins);
    const float sfq = powf(sqrtf(mean), 0.8f);
    return floor_ > (double)sfq ? floor_ * loss : (double)(sfq * (float)loss);
}

// =============================================================================================
// encode: PCM -> payload.  grid = min(ceil(units / 7), CUs), block = 448 (7 independent waves).
// unit = one frame (CC == 2) or a pair of frames 2u, 2u+1 (CC == 1).
// Requires: pcm 16-byte aligned, frame byte stride % 16 == 0, payload 16-byte aligned, payload_stride % 16 == 0.
//
// The wave's 17 KiB of LDS, by phase:
//   [8 KiB, 16 KiB)  raw PCM of the unit, brought in by LDS-DMA (global_load_lds_dwordx4: no registers) -- 1- and 2-byte
//                    samples.  Issued for the NEXT unit as soon as the exchange has been consumed, so it lands during
//                    pass 2, the pair step and the stores.  4- and 8-byte samples are loaded per element at the top of
//                    the unit instead (the SIMD's other wave covers the latency).
//   [0, 17 KiB)      the two exchange planes (after the raw bytes have been read into registers)
//   [0, 8 KiB)       payload staging, two buffers of four 1 KiB rows: the pair jobs run in NB groups, each of which
//                    completes four 1 KiB pieces of the payload (bins [g B, (g+1) B), their mirror images below M,
//                    the same above M, and the mirror below N; B = 512 / NB bins); the pieces go to LDS in payload
//                    byte order, come back as 16 bytes per lane and leave as four coalesced 1 KiB stores -- stores are
//                    spread over the whole pair step instead of ending the unit in one burst.
// Every table / LDS read is issued a batch ahead of its use (explicit double buffering between scheduling fences):
// with two waves per SIMD an LDS round trip (hundreds of cycles behind the other waves' bursts) is otherwise exposed.
// =============================================================================================
#define FRAD_FENCE() __builtin_amdgcn_sched_barrier(0)

// sums over the two 32-lane halves of a wave (all lanes get both)
__device__ __forceinline__ void half_wave_sum_f64(double v, double& lo, double& hi) {
#ifdef FRAD_HOST_EMULATION
    for (int off = 1; off < 32; off <<= 1) v = v + u2d(__shfl_xor(d2u(v), off, 64));
    const double other = u2d(__shfl_xor(d2u(v), 32, 64));
    const bool up = (threadIdx.x & 32) != 0;
    lo = up ? other : v; hi = up ? v : other;
#else
    v = v + u2d(dpp_move_u64<0xB1>(d2u(v)));
    v = v + u2d(dpp_move_u64<0x4E>(d2u(v)));
    v = v + u2d(dpp_move_u64<0x141>(d2u(v)));
    v = v + u2d(dpp_move_u64<0x140>(d2u(v)));
    lo = u2d(read_lane_u64(d2u(v), 0)) + u2d(read_lane_u64(d2u(v), 16));
    hi = u2d(read_lane_u64(d2u(v), 32)) + u2d(read_lane_u64(d2u(v), 48));
#endif
}
__device__ __forceinline__ int wave_read_lane(int v, int src) {      // v of lane `src` (wave-uniform index), as a scalar
#ifdef FRAD_HOST_EMULATION
    return (int)__shfl((unsigned long long)(unsigned)v, src, 64);
#else
    return __builtin_amdgcn_readlane(v, src);
#endif
}
__device__ __forceinline__ bool wave_any(bool b) {                     // true in some lane (all 64 lanes call it)
#ifdef FRAD_HOST_EMULATION
    unsigned long long v = b ? 1 : 0;
    for (int off = 1; off < 64; off <<= 1) v |= __shfl_xor(v, off, 64);
    return v != 0;
#else
    return __builtin_amdgcn_ballot_w64(b) != 0;
#endif
}
__device__ __forceinline__ int wave_uniform_int(int v) {
#ifdef FRAD_HOST_EMULATION
    return (int)__shfl((unsigned long long)(unsigned)v, 0, 64);
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// ---- K7 tail (round 3): Z (E, O as pass 2 leaves them) -> quantised integers ---------------------------------------
// profile1.py:21-40, p1tools.py:15-44.  The DCT pair step runs ONCE and leaves the frame's 4096 coefficients in the
// registers that held E / O (lane 0's different pairing is resolved by one permutation up front, so all 16 jobs read the
// same registers in every lane).  Band energies then need "lane owns consecutive bins", the quantiser "a wave store covers
// consecutive bins"; the same registers serve both through one trip over the wave's LDS buffer:
//   pass A  coefficients -> LDS in bin order (two rounds of 1024 bins x both channels: 2 x 32 rows of 32 doubles, rows
//           padded to 34 as in `pslot`), read back as rows: lane (h, r) owns bins [1024 R + 32 r, + 32) of channel h.  A
//           run of 32 bins touches at most three bands (launch condition), so three exec-masked running sums per lane
//           replace the ~110 masked half-wave reductions of the round-2 kernel; the partial sums meet in a small
//           [band][run] table and lane b (< 27) of each half adds band b's runs in order (deterministic).
//   thresholds, their integer codes and the ramp steps: one band per lane (as before).
//   pass B  per bin: band and position inside the band from two LDS byte / short tables (the bin index is lane + constant,
//           so the reads need no address arithmetic), {threshold, step} of that band, numpy's linspace arithmetic, the
//           float32-decided quantiser.  The integers go to the staging rows of the profile-0 kernel (32-bit codes) and
//           leave as 16 coalesced 1 KiB stores.
// The next unit's PCM DMA is issued between the two passes (pass A needs the whole 17 KiB buffer).
// numpy's linspace has a second branch for a step that underflows to zero while the end points differ (p1tools.py:35-41
// -> numpy.linspace `any_step_zero`).  It cannot be reached here: a threshold is either 0 (bands past the first empty
// one) or >= min(ATH, 1) x loss >= 0.56 x 0.125, so two thresholds differ by 0, by one of them, or by >= one ulp of
// 0.07 ~ 1.4e-17, and the band is at most 2048 bins wide: the quotient never underflows.
struct P1K7Lds { const unsigned char* band; const int* edge; const double* floor_; const unsigned short* pk; const double* tqh; };
__device__ __forceinline__ P1K7Lds p1w_k7_lds(unsigned char* smem) {
    unsigned char* b = smem + kWaveLdsBytes + 16;
    return {b, reinterpret_cast<const int*>(b + 2048), reinterpret_cast<const double*>(b + 2048 + 128), reinterpret_cast<const unsigned short*>(b + 2048 + 128 + 256),
            reinterpret_cast<const double*>(b + 2048 + 128 + 256 + 4096)};
}
// float32 hardware transcendentals (one instruction each, about 1 ulp): seeds and decisions only, never a stored value
__device__ __forceinline__ float k7_log2f(float v) {
#ifdef FRAD_HOST_EMULATION
    return log2f(v);
#else
    return __builtin_amdgcn_logf(v);
#endif
}
__device__ __forceinline__ float k7_exp2f(float v) {
#ifdef FRAD_HOST_EMULATION
    return exp2f(v);
#else
    return __builtin_amdgcn_exp2f(v);
#endif
}
__device__ __forceinline__ float k7_rcpf(float v) {
#ifdef FRAD_HOST_EMULATION
    return 1.0f / v;
#else
    return __builtin_amdgcn_rcpf(v);
#endif
}
// r^0.4 = sqrt(r)^0.8 (p1tools.py:30, r = mean of the squared band), float64 to a few ulp without libm (whose constants the
// compiler hoists out of the frame loop into registers it then spills): float32 seed, two Newton steps on y^5 = r^2, the
// division inside a step by a float32 reciprocal (it scales a correction of relative size 1e-5 and 1e-11).
__device__ __forceinline__ double k7_pow04(double r) {
    const float rf = (float)r;
    if (!(rf > 1e-30f)) return 0.0;                           // (far below every absolute threshold of hearing: the floor wins)
    double y = (double)k7_exp2f(0.4f * k7_log2f(rf));
    const double r2 = r * r;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
        const double corr = (r2 - y5) * (double)k7_rcpf((float)y5);
        y = fma(y * 0.2, corr, y);
    }
    return y;
}
// (int) round(dequant(log(max(t, 1)) / log(e / 2))) (profile1.py:38-40) = the number of table entries tqh[n] <= t: float32
// estimate, then one exact step either way.  Beyond the table (band RMS above 3e8 in scaled units: not reachable with PCM of
// 32 bits or fewer at the loss levels the reference offers) the float32 estimate stands.
__device__ __forceinline__ int32_t k7_band_code(double t, const double* tqh) {
    if (!(t > 1.0)) return 0;
    const float v = k7_log2f((float)t) * 2.2585796f;          // ln 2 / ln(e / 2)
    const float w = k7_exp2f(k7_log2f(v) * (4.0f / 3.0f));
    if (!(w < 255.4f)) return w < 2.1e9f ? (int32_t)rintf(w) : 0x7fffffff;
    int n = (int)rintf(w);
    if (n > 0 && t < tqh[n - 1]) n -= 1;
    else if (t >= tqh[n]) n += 1;
    return n;
}
// (int) round(sign(x) |x / div|^0.75), x = X scale (the pair step's tables carry the scale): the per-bin quantiser of profile1.py:27-36.  Decided in float32 wherever that is
// safe: the float32 value of |.|^0.75 carries a relative error below 1e-6 (two conversions, a reciprocal, two square roots,
// three products), so unless it lies within 4e-6 y of a half-integer it rounds like the exact value.  The undecided bin (about
// 1e-5 y of them) refines the float32 value by two Newton steps on y^4 div^3 = (|x| scale)^3 -- no division, no libm -- and,
// should that still sit within 1e-12 of a half-integer h, compares (|x| scale)^3 with h^4 div^3 directly.
template <bool EXACT>
__device__ __forceinline__ int32_t k7_quantise(double x, double div, bool& undecided) {      // x = X 2^(bits-1) already
    const float xf = (float)x, df = (float)div;
    const float mf = fabsf(xf) * k7_rcpf(df);
    const float r = p1w_sqrtf(mf), yf = r * p1w_sqrtf(r);
    const float ys = copysignf(yf, xf), rs = rintf(ys);       // the sign goes in before the rounding (round-half-even is symmetric)
    int32_t qa = (int32_t)rs;
    const bool und = !(fabsf(ys - rs) < 0.5f - (yf * 4e-6f + 1e-6f));      // (also NaN / Inf: div == 0, non-finite input)
    if constexpr (!EXACT) undecided |= und;
    else if (und) {
        if (div == 0.0 || x == 0.0 || x != x || div != div) qa = 0;              // x / inf -> 0; NaN -> 0 like the conversion
        else if (!(mf < 1e30f)) qa = xf < 0.0f ? -0x7fffffff : 0x7fffffff;       // beyond int32 (the exact conversion saturates as well)
        else {
            const double a = fabs(x), a3 = a * a * a, d3 = div * div * div;
            double y = (double)yf;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double y2 = y * y, den = y2 * y2 * d3;
                y = fma(y * 0.25, (a3 - den) * (double)k7_rcpf((float)den), y);
            }
            const double n = floor(y), hh = n + 0.5;
            const double up = fabs(y - hh) > y * 1e-12 ? (y > hh ? 1.0 : 0.0) : (a3 > (hh * hh) * (hh * hh) * d3 ? 1.0 : 0.0);
            const double qd = n + up;
            qa = qd < 2147483647.0 ? (int32_t)qd : 0x7fffffff;
            qa = xf < 0.0f ? -qa : qa;
        }
    }
    return qa;
}
// pk[(s * 32 + l) * 4 + cls] = (band j of the bin that lane l holds for job slot s, class cls) << 10 | position of the bin inside
// band j; classes: X[k], X[M - k], X[M + k], X[N - k], k = wave_job_k(l, s) (lane 0 of slot 0: bins 0, 512, 1024, 1536).  One 8-byte
// read per lane and job; both halves of the wave read the same entries.  After p1w_tables_to_lds + a block barrier.
__device__ __forceinline__ int k7_bin_of(int s, int cls, int l) {
    const int k = wave_job_k(l, s);
    if (s == 0 && l == 0) return 512 * cls;
    return cls == 0 ? k : cls == 1 ? 1024 - k : cls == 2 ? 1024 + k : 2048 - k;
}
__device__ __forceinline__ void p1w_k7_tables(unsigned char* smem) {
    unsigned char* b = smem + kWaveLdsBytes + 16;
    const int* edge = reinterpret_cast<const int*>(b + 2048);
    unsigned short* pk = reinterpret_cast<unsigned short*>(b + 2048 + 128 + 256);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const int cls = i & 3, l = (i >> 2) & 31, s = i >> 7;
        const int kb = k7_bin_of(s, cls, l), j = b[kb];
        pk[i] = (unsigned short)(j < 26 ? ((kb - edge[j]) | (j << 10)) : 0);
    }
}
constexpr int kK7PlaneBytes = 32 * 34 * 8;            // one channel's 1024 coefficients of a round (8704 B; two planes = the wave's buffer)
constexpr int kK7RecOff = 16384;                      // {threshold, step} per band and half: 2 x 32 x 16 B, above the PCM landing zone
constexpr int kK7Slots = 16;                          // runs of 32 bins per band in the gather table (launch condition)
static_assert(2 * kK7PlaneBytes <= kWaveBufBytes && kK7RecOff + 1024 <= kWaveBufBytes && 2 * 27 * kK7Slots * 8 <= 8192, "K7 LDS plan");

template <int CC, typename T, typename DMA, typename STAMP>
__device__ __forceinline__ void wave_p1_tail(cx<T> (&E)[16], cx<T> (&O)[16], const cx<T>* ltab, unsigned char* smem,
                                             unsigned char* wbuf, unsigned char* __restrict__ qout, const P1Wave& pw, const Geom& g,
                                             long long u, int lane, DMA&& dma_next, STAMP&& stamp) {
    constexpr int M = 1024, N = 2048;
    const int h = lane >> 5, l = lane & 31;
    const bool lane0 = (l == 0);
    const long long f = CC == 2 ? u : 2 * u + h;
    const bool live = f < g.n_frames;
    const int c = CC == 2 ? h : 0;
    const P1K7Lds t = p1w_k7_lds(smem);
    FRAD_MARK("k7_pair_step");
    auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
    // ---- the self-paired bin k = 512 (lane 0's E[8]) and lane 0's pairing --------------------------------------------
    T xs0, xs1;
    {
        const cx<T> zk = E[8], zp = conj(E[8]);
        const cx<T> p = cmul(zk + zp, ltab[WaveLayout::TW1 + 0]), qq = cmul(zk - zp, ltab[WaveLayout::TW1 + 1]);
        const cx<T> S = p + qq;
        xs0 = S.x; xs1 = -S.y;
    }
    // job s reads zk = (s < 8 ? E[s] : O[15 - s]) and zm = (s < 8 ? O[15 - s] : E[s]); lane 0 pairs E[s] with E[16 - s] and
    // O[15 - s] with O[s] instead (see wave_fwd_body), i.e. it wants E[8 .. 15] <- O[8 .. 15] and O[j] <- E[(j + 1) & 15], j >= 8
#pragma unroll
    for (int j = 8; j < 16; ++j) {                            // ascending: E[j + 1] is still the old value when O[j] takes it
        const cx<T> eo = E[j], oo = O[j];
        E[j] = sel(oo, eo);
        O[j] = sel(E[(j + 1) & 15], oo);
    }
    // ---- pair step, once: X[s][0..3] = X[k], X[M - k], X[M + k], X[N - k], k = wave_job_k(l, s); lane 0 of slot 0 (k = 0)
    //      carries X[0], X[512], X[M], X[1536] (the self-paired bins take the places of the two bins it does not have)
    // Registers hold ONE half of the coefficients at a time while the band energies are taken (XH = classes 2, 3: the bins from M
    // up; XL = classes 0, 1), the wave's planes the other: all 128 registers' worth live through the pair step and pass A had 25
    // doubles per lane spilled to scratch -- 12.8 KB per frame that did reach HBM (578 MB measured for 372 MB algorithmic).
    T XH[16][2], XL[16][2];
    // plane address of local bin b (0 .. 1023): row b >> 5 (pitch 34 doubles), column b & 31
    auto paddr = [&](int b) -> int { return ((b >> 5) * 34 + (b & 31)) * 8; };
    unsigned char* plw = wbuf + h * kK7PlaneBytes;
    {
        cx<T> ptab[2][2];
        auto ptab_load = [&](int s) {
            ptab[s & 1][0] = ltab[WaveLayout::PW + s * 32 + l];
            ptab[s & 1][1] = ltab[WaveLayout::PG + s * 32 + l];
        };
        ptab_load(0);
        FRAD_FENCE();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) ptab_load(s + 1);
            FRAD_FENCE();
            const cx<T> zk = s < 8 ? E[s] : O[15 - s];
            const cx<T> zm = s < 8 ? O[15 - s] : E[s];
            const cx<T> zp = conj(zm);
            const cx<T> p = cmul(zk + zp, ptab[s & 1][0]), qq = cmul(zk - zp, ptab[s & 1][1]);
            const cx<T> S = p + qq, D = p - qq;
            const T xm = (D.x - D.y) * K<T>::s2, xp = (D.x + D.y) * K<T>::s2;
            T x0, x1;
            if (s == 0) { x0 = S.x; x1 = lane0 ? xs0 : xm; XH[0][0] = lane0 ? xm : xp; XH[0][1] = lane0 ? xs1 : -S.y; }
            else { x0 = S.x; x1 = xm; XH[s][0] = xp; XH[s][1] = -S.y; }
            // round 0 of pass A: bins below M go to the plane as they appear (and stay only there until round 1)
            const int k = wave_job_k(l, s);
            const int kb1 = (s == 0 && lane0) ? 512 : M - k;
            *reinterpret_cast<T*>(plw + paddr(k)) = x0;
            *reinterpret_cast<T*>(plw + paddr(kb1)) = x1;
            FRAD_FENCE();
        }
    }
    stamp(ic<7>{}); FRAD_MARK("k7_band_sums");
    // ---- pass A: band energies --------------------------------------------------------------------------------------
    // per-launch constants of this lane's two runs (g = 32 R + l): first band, offsets of the (at most two) band edges inside
    // the run (32 = none), slot of the run in its first band's table row
    int rb0[2], rr1[2], rr2[2], rsl[2];
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        const int gq = 32 * R + l, b0 = t.band[32 * gq];
        const int e1 = t.edge[b0 + 1] - 32 * gq, e2 = t.edge[b0 + 2] - 32 * gq;     // (edge[] is clipped to N and padded with N)
        rb0[R] = b0; rr1[R] = e1 < 32 ? e1 : 32; rr2[R] = e2 < 32 ? e2 : 32; rsl[R] = gq - (t.edge[b0] >> 5);
    }
    double part[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    auto run_sums = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        const v2d* row = reinterpret_cast<const v2d*>(plw + l * 272);
        const int r1 = rr1[R], r2 = rr2[R];
        double a = 0.0, b = 0.0, cc = 0.0;
        v2d buf[2][4];
        auto fetch = [&](int bq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) buf[bq & 1][i] = row[4 * bq + i];
        };
        fetch(0);
        FRAD_FENCE();
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            if (bq < 3) fetch(bq + 1);
            FRAD_FENCE();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int idx = 8 * bq + 2 * i + e;
                    const double x = buf[bq & 1][i][e];
                    if (idx < r1) a = fma(x, x, a); else if (idx < r2) b = fma(x, x, b); else cc = fma(x, x, cc);
                }
            }
            FRAD_FENCE();
        }
        part[R][0] = a; part[R][1] = b; part[R][2] = cc;
    };
    team_sync<64>();
    run_sums(ic<0>{});
    team_sync<64>();                                              // round 0 read: the planes take the upper half of the bins
#pragma unroll
    for (int s = 0; s < 16; ++s) {                                // the halves change places: a lane reads back exactly what it wrote
        const int k = wave_job_k(l, s);
        const int kb3 = (s == 0 && lane0) ? 512 : M - k;          // local bin of N - k (lane 0, slot 0: 1536)
        T* pa = reinterpret_cast<T*>(plw + paddr(k));
        T* pb = reinterpret_cast<T*>(plw + paddr(kb3));
        XL[s][0] = *pa; XL[s][1] = *pb;
        *pa = XH[s][0]; *pb = XH[s][1];
    }
    team_sync<64>();
    run_sums(ic<1>{});
    team_sync<64>();
#pragma unroll
    for (int s = 0; s < 16; ++s) {                                // the upper half back into registers: the buffer is about to be reused
        const int k = wave_job_k(l, s);
        const int kb3 = (s == 0 && lane0) ? 512 : M - k;
        XH[s][0] = *reinterpret_cast<const T*>(plw + paddr(k));
        XH[s][1] = *reinterpret_cast<const T*>(plw + paddr(kb3));
    }
#ifndef FRAD_HOST_EMULATION
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every plane read has returned: the DMA may land in [8 KiB, 16 KiB)
#endif
    team_sync<64>();
    FRAD_FENCE();
    dma_next();
    FRAD_FENCE();
    stamp(ic<8>{}); FRAD_MARK("k7_thresholds");
    // partial sums -> table [half][band][run slot]; lane b adds band b's runs
    double* tab = reinterpret_cast<double*>(wbuf) + h * (27 * kK7Slots);
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        tab[rb0[R] * kK7Slots + rsl[R]] = part[R][0];
        if (rr1[R] < 32) tab[(rb0[R] + 1) * kK7Slots] = part[R][1];
        if (rr2[R] < 32) tab[(rb0[R] + 2) * kK7Slots] = part[R][2];
    }
    team_sync<64>();
    double* rec = reinterpret_cast<double*>(wbuf + kK7RecOff) + h * 64;     // rec[2 b] = threshold, rec[2 b + 1] = ramp step
    if (l < 27) {
        const int b = l, e0 = t.edge[b], e1 = t.edge[b + 1], bins = e1 - e0;
        const int cnt = bins > 0 ? ((e1 - 1) >> 5) - (e0 >> 5) + 1 : 0;
        const v2d* tr = reinterpret_cast<const v2d*>(tab + b * kK7Slots);
        double energy = 0.0;
#pragma unroll
        for (int i = 0; i < kK7Slots / 2; ++i) {
            const v2d v = tr[i];
            energy += (2 * i < cnt) ? v[0] : 0.0;
            energy += (2 * i + 1 < cnt) ? v[1] : 0.0;
        }
        double th = 0.0;
        if (b < pw.nb_used) {                                     // p1tools.py:30-31
            const double sfq = k7_pow04(energy / (double)bins), fl = t.floor_[b];
            th = (fl > sfq ? fl : sfq) * pw.loss;
        }
        rec[2 * b] = th;
        if (live) pw.tq_out[(f * 27 + b) * CC + c] = k7_band_code(th, t.tqh);
    }
    team_sync<64>();
    if (l < 27) {
        double st = 0.0;
        if (l < 26) { const int num = t.edge[l + 1] - t.edge[l]; if (num > 0) st = (rec[2 * l + 2] - rec[2 * l]) / (double)num; }
        rec[2 * l + 1] = st;
    }
    team_sync<64>();
    stamp(ic<9>{}); FRAD_MARK("k7_quantiser");
    // ---- pass B: quantiser (profile1.py:27-36), integers through the profile-0 staging rows ---------------------------
    constexpr int NB = 4, NG = 4, JPG = 4, BPC = 128, ES = CC * NB;
    int hq = h, lq = l;
    FRAD_OPAQUE(hq); FRAD_OPAQUE(lq);
    unsigned char* stg = wbuf + (CC == 2 ? hq * NB : hq * 512);
    unsigned char* dstf = qout + (live ? f : 0) * g.payload_stride + (CC == 2 ? lane : l) * 16;
    const int la = lq * ES, lb = (lane0 ? 32 : 64 - lq) * ES;
    auto row_of = [&](int gi, int cls) -> unsigned char* { return stg + (gi & 1) * 4096 + cls * 1024; };
    auto job_slot = [](int gi, int i) -> int { const int tt = gi * (JPG / 2) + (i >> 1); return (i & 1) ? 15 - tt : tt; };
    const v2d* recv = reinterpret_cast<const v2d*>(rec);
    const v2u* pk2 = reinterpret_cast<const v2u*>(t.pk) + l;      // entry of (slot s, this lane): pk2[32 s]
    auto code_store = [&](unsigned char* p, int32_t v) { *reinterpret_cast<int32_t*>(p) = v; };
    // The look-ups of a job -- its pk entry, then the {threshold, step} records of its four bins -- are two dependent LDS round
    // trips; they run one job (records) and two jobs (entry) ahead of the arithmetic, across the groups.  Job j = 0 .. 15 is
    // (group NG - 1 - j / JPG, place j % JPG).
    auto slot_of_job = [&](int j) -> int { return job_slot(NG - 1 - j / JPG, j % JPG); };
    v2u ent_cur = pk2[32 * slot_of_job(0)], ent_nxt = pk2[32 * slot_of_job(1)];
    v2d rc_cur[4];
    {
        const uint32_t en0[4] = {ent_cur[0] & 0xffffu, ent_cur[0] >> 16, ent_cur[1] & 0xffffu, ent_cur[1] >> 16};
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) rc_cur[cq] = recv[en0[cq] >> 10];
    }
    FRAD_FENCE();
#pragma unroll
    for (int gi = NG - 1; gi >= 0; --gi) {
        bool undecided = false;
        // the group's four jobs; EXACT = false: float32 decisions, `undecided` collects the bins that float32 cannot decide;
        // EXACT = true (run only when some lane has one): every bin again with the exact fall-back, same stores
        auto group = [&](auto exact_tag) {
            constexpr bool EXACT = decltype(exact_tag)::value;
#pragma unroll
            for (int i = 0; i < JPG; ++i) {
                const int s = job_slot(gi, i), j = (NG - 1 - gi) * JPG + i;
                v2u e2; v2d rc[4];
                v2u ent_nn = ent_nxt; v2d rc_nxt[4];
                if constexpr (EXACT) {                            // the rare second run of a group fetches for itself
                    e2 = pk2[32 * s];
                    const uint32_t ex[4] = {e2[0] & 0xffffu, e2[0] >> 16, e2[1] & 0xffffu, e2[1] >> 16};
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) rc[cq] = recv[ex[cq] >> 10];
                } else {
                    e2 = ent_cur;
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) rc[cq] = rc_cur[cq];
                    if (j + 1 < 16) {                             // next job's records (its entry arrived a job ago), the entry after that
                        const uint32_t en1[4] = {ent_nxt[0] & 0xffffu, ent_nxt[0] >> 16, ent_nxt[1] & 0xffffu, ent_nxt[1] >> 16};
#pragma unroll
                        for (int cq = 0; cq < 4; ++cq) rc_nxt[cq] = recv[en1[cq] >> 10];
                        if (j + 2 < 16) ent_nn = pk2[32 * slot_of_job(j + 2)];
                    }
                }
                const uint32_t en[4] = {e2[0] & 0xffffu, e2[0] >> 16, e2[1] & 0xffffu, e2[1] >> 16};
                FRAD_FENCE();
                int32_t qv[4];
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) {                  // np.linspace without its end point: t0 + i * step (two roundings)
                    const double y = (double)(int)(en[cq] & 1023u) * rc[cq][1];
                    qv[cq] = k7_quantise<EXACT>(cq < 2 ? XL[s][cq] : XH[s][cq - 2], y + rc[cq][0], undecided);
                }
                const int tt = s < 8 ? s : 15 - s, tl = tt - gi * (JPG / 2);
                const int offa = 64 * tl * ES + (s < 8 ? la : lb);
                unsigned char* pa = row_of(gi, 0) + offa;
                unsigned char* pc = row_of(gi, 2) + offa;
                unsigned char* pb = row_of(gi, 1) + BPC * ES - offa;
                unsigned char* pd = row_of(gi, 3) + BPC * ES - offa;
                code_store(pa, qv[0]);
                code_store(pc, qv[2]);
                if (s == 0) {
                    // lane 0 holds k = 0: X[0] (class A) and X[M] (first bin of class C); its bins 512 / 1536 were stored with the
                    // rows of group NG - 1 (below); here it repeats X[0]
                    code_store(lane0 ? pa : pb, lane0 ? qv[0] : qv[1]);
                    code_store(lane0 ? pa : pd, lane0 ? qv[0] : qv[3]);
                } else if (tl == 0 && s < 8) {                    // lane 0's M - k and N - k open the next-higher piece (group gi - 1)
                    code_store(lane0 ? row_of(gi - 1, 1) : pb, qv[1]);
                    code_store(lane0 ? row_of(gi - 1, 3) : pd, qv[3]);
                } else {
                    code_store(pb, qv[1]);
                    code_store(pd, qv[3]);
                }
                if constexpr (!EXACT) {
                    if (j + 1 < 16) {
                        ent_cur = ent_nxt; ent_nxt = ent_nn;
#pragma unroll
                        for (int cq = 0; cq < 4; ++cq) rc_cur[cq] = rc_nxt[cq];
                    }
                }
                FRAD_FENCE();
            }
            if (gi == NG - 1) {                                   // lane 0: bins 512 and 1536 (its slot-0 values of classes 1 and 3)
                const v2u e2 = pk2[0];
                const uint32_t e1 = e2[0] >> 16, e3 = e2[1] >> 16;
                const v2d r1 = recv[e1 >> 10], r3 = recv[e3 >> 10];
                const int32_t q1 = k7_quantise<EXACT>(XL[0][1], (double)(int)(e1 & 1023u) * r1[1] + r1[0], undecided);
                const int32_t q3 = k7_quantise<EXACT>(XH[0][1], (double)(int)(e3 & 1023u) * r3[1] + r3[0], undecided);
                if (lane0) { code_store(row_of(NG - 1, 1), q1); code_store(row_of(NG - 1, 3), q3); }
            }
        };
        group(ic<0>{});
        if (wave_any(undecided)) group(ic<1>{});
        FRAD_FENCE();
        team_sync<64>();
        FRAD_FENCE();
        {
            v4u row[4];
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) row[cq] = *reinterpret_cast<const v4u*>(wbuf + (gi & 1) * 4096 + cq * 1024 + lane * 16);
            team_sync<64>();
            const int oa = gi * 512 * CC, ob = 1024 * ES - (gi + 1) * 512 * CC, oc = 1024 * ES + gi * 512 * CC, od = 2048 * ES - (gi + 1) * 512 * CC;
            const int off[4] = {oa, ob, oc, od};
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
                if (CC == 2 || live) stream_store(dstf + off[cq], row[cq]);
        }
        FRAD_FENCE();
    }
}
#ifndef FRAD_WAVE_DMA_AUX
#define FRAD_WAVE_DMA_AUX 2                     // nt: the PCM is read once
#endif
#ifndef FRAD_WAVE_TWB
#define FRAD_WAVE_TWB 4
#endif
#ifndef FRAD_WAVE_JOBFENCE
#define FRAD_WAVE_JOBFENCE 1
#endif
#ifndef FRAD_WAVE_PRB
#define FRAD_WAVE_PRB 4
#endif

// CLIPS: the batch is a set of equally cut clips (Geom::fpc, clip_stride; frad_p0_analogue_clips) -- a variant of its own, so
// that the flat batch keeps its frame stride in one scalar product (the clip arithmetic cost the headline kernel its registers)
template <int LG, int CC, int BITS, int MODE, typename P1, bool CLIPS = false>
__device__ __forceinline__ void
wave_fwd_body(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<double>* __restrict__ blob, const Geom& g, const P1& pw) {
    using T = double;
    constexpr int M = 1024, N = 2048, ISZ = 1 << LG, NB = BITS / 8;
    static_assert(BITS == 16 || BITS == 32 || BITS == 64, "whole-byte power-of-two storage");
    static_assert(CC == 1 || CC == 2, "channels");
    constexpr int RAWB = N * ISZ * CC;                 // raw bytes of one frame
    constexpr int FPW = 2 / CC;                        // frames per wave
    constexpr int QB = 4 * CC * ISZ;                   // bytes of four consecutive sample-frames
    constexpr bool STAGED = RAWB * FPW <= 8192;        // raw bytes through LDS (else: per-element global loads)
    constexpr int RAWOFF = 8192;                       // where the raw bytes land in the wave's buffer
    constexpr int NDMA = STAGED ? RAWB * FPW / 1024 : 0;   // 1 KiB LDS-DMA pieces per unit
    constexpr int NG = NB;                             // groups of pair jobs = payload pieces per bin class
    constexpr int JPG = 16 / NG;                       // jobs per group
    constexpr int BPC = 512 / NB;                      // bins per 1 KiB piece (per frame: 512 CC bytes)
    constexpr int ES = CC * NB;                        // payload bytes per bin
    FRAD_DYN_SMEM(smem);
    const T deferred = (T)pcm_deferred_scale(g.dtype, g.raw_be);
    T sc = ((T)1 / (T)(2 * N)) * deferred;                   // exact power of two
    if constexpr (MODE == 1) sc *= (T)pw.scale;               // profile 1: the coefficients leave the pair step as X 2^(bits-1) (profile1.py:26-27:
                                                              //  every later use has that factor; a power of two commutes with the roundings)
    {
        cx<T>* l = reinterpret_cast<cx<T>*>(smem);
        for (int i = threadIdx.x; i < WaveLayout::SLOTS; i += blockDim.x) {
            cx<T> v = blob[i];
            if (i >= WaveLayout::PW || i < 2) { v.x *= sc; v.y *= sc; }
            l[i] = v;
        }
    }
    const cx<T>* ltab = reinterpret_cast<const cx<T>*>(smem);
    if constexpr (MODE == 1) p1w_tables_to_lds(smem, pw);
#ifdef FRAD_HOST_EMULATION
    const int wv = threadIdx.x >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform: unit numbers and frame bases stay in SGPRs
#endif
    unsigned char* wbuf = smem + kWaveTableBytes + wv * kWaveBufBytes;
    const long long frameb = (g.frame_stride * CC) << LG;
    [[maybe_unused]] const long long gapb = ((g.clip_stride - (long long)g.fpc * g.frame_stride) * CC) << LG;   // bytes skipped between clips
    auto frame_off = [&](long long f) -> long long {          // byte offset of frame f in the PCM buffer
        if constexpr (CLIPS) return f * frameb + (long long)((unsigned)f / (unsigned)g.fpc) * gapb; else return f * frameb;
    };
    const long long n_units = (g.n_frames + FPW - 1) / FPW;
    // units [ub, ue) belong to this block; its waves draw them from an LDS counter (the first kWaveWaves statically)
    const long long ub = n_units * blockIdx.x / gridDim.x, ue = n_units * (blockIdx.x + 1) / gridDim.x;
    unsigned* ctr = reinterpret_cast<unsigned*>(smem + kWaveTableBytes + kWaveWaves * kWaveBufBytes);
    if (threadIdx.x == 0) *ctr = (unsigned)kWaveWaves;
    // byte-order of the stored codes as a v_perm_b32 selector (one instruction whatever the endianness)
    const uint32_t psel = g.le ? 0x07060504u : (BITS == 16 ? 0x07070405u : 0x04050607u);
    int lane = threadIdx.x & 63;

    auto frame_of = [&](long long u, int h) -> long long {   // frame this lane works on (clamped for the odd mono tail)
        if constexpr (CC == 2) return u;
        else { const long long f = 2 * u + h; return f < g.n_frames ? f : g.n_frames - 1; }
    };
    // raw bytes of unit u -> LDS.  Piece i is one wave instruction: lane t's 16 bytes land at RAWOFF + 1024 i + 16 t.
    // CC == 2: the frame's bytes in order.  CC == 1: lanes 0-31 carry frame 2u, lanes 32-63 frame 2u+1, 512 bytes each
    // per piece, i.e. byte b of half h's frame lives at 1024 (b >> 9) + 512 h + (b & 511).
    auto dma_in = [&](long long u) {
        if constexpr (STAGED) {
            const int h = lane >> 5, l = lane & 31;
            const unsigned char* src = pcm + frame_off(frame_of(u, h)) + (CC == 2 ? lane : l) * 16;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
#ifdef FRAD_HOST_EMULATION
                *reinterpret_cast<v4u*>(wbuf + RAWOFF + i * 1024 + lane * 16) = *reinterpret_cast<const v4u*>(src + i * (CC == 2 ? 1024 : 512));
#else
#ifndef FRAD_X_NODMA
                __builtin_amdgcn_global_load_lds(FRAD_GCPTR(void, src + i * (CC == 2 ? 1024 : 512)),
                                                 (__attribute__((address_space(3))) void*)(wbuf + RAWOFF + i * 1024), 16, 0, FRAD_WAVE_DMA_AUX);
#endif
#endif
            }
        }
    };
    auto raw_at = [&](int h, int b) -> int {                  // LDS offset of byte b (a compile-time multiple of QB plus a lane term) of half h's frame
        if constexpr (CC == 2) return RAWOFF + b;
        else return RAWOFF + 1024 * (b >> 9) + 512 * h + (b & 511);
    };

    long long u = ub + wv;
    if (u < ue) dma_in(u);
    __syncthreads();                                          // tables and counter are in LDS (the barrier's fence also retires the first DMA)
    if constexpr (MODE == 1) { p1w_k7_tables(smem); __syncthreads(); }
    for (int i = (wv * 8 + (int)(blockIdx.x & 7)) * g.cg; i > 0; --i) FRAD_WAVE_SLEEP(1);      // start stagger (g.cg x 64 cycles per step; 0 = off)
    FRAD_STAMP_DECL;
    while (u < ue) {
        lane = threadIdx.x & 63; FRAD_OPAQUE(lane);           // per-lane addresses are rebuilt each unit (no LICM register hoard)
        const int h = lane >> 5, l = lane & 31;
        const long long next = ub + wave_next_unit(ctr);      // (needed half a unit later, for the DMA)
        cx<T> z[32];                                          // z[j] = packed sequence point l + 32 j of this lane's channel
        const cx<T>* tw = ltab + WaveLayout::TW1 + l;
        constexpr int TWB = FRAD_WAVE_TWB;                    // rows of TW1 per twiddle batch (double buffered)
        cx<T> twb[2][TWB];
        auto tw_load = [&](int b) {
#pragma unroll
#ifdef FRAD_X_NOTW
            for (int i = 0; i < TWB; ++i) twb[b & 1][i] = cx<T>{sc, deferred};
#else
            for (int i = 0; i < TWB; ++i) if (TWB * b + i > 0) twb[b & 1][i] = tw[(TWB * b + i) * 32];
#endif
        };
        if constexpr (STAGED) {
#ifndef FRAD_HOST_EMULATION
            // this unit's DMA was issued before the previous unit's 4 NB row stores: vector-memory operations retire
            // in order, so at most that many may still be in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(4 * NB) : "memory");   // (profile 1: the 16 row stores follow the DMA, and one store of threshold codes before them)
#endif
            FRAD_STAMP(0); FRAD_MARK("makhoul_convert");
            team_sync<64>();
#ifdef FRAD_X_NOCOMPUTE
            {
                dma_in(next < ue ? next : u);
                const long long f = CC == 2 ? u : 2 * u + h;
                unsigned char* dstf = payload + f * g.payload_stride + lane * 16;
#pragma unroll
                for (int gi = 0; gi < 16; ++gi) {
                    const v4u row = *reinterpret_cast<const v4u*>(wbuf + RAWOFF + (gi & 7) * 1024 + lane * 16);
                    *FRAD_GPTR(v4u, dstf + gi * 1024) = row;
                }
                team_sync<64>();
                u = next;
                continue;
            }
#endif
            // ---- Makhoul pairs: whole quads (consecutive lanes read consecutive QB-byte groups) ----
            dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
                constexpr int CODE = decltype(code_tag)::value;
                constexpr bool RAW = decltype(raw_tag)::value != 0;
                constexpr int NW = QB / 4, EB = 8 * ISZ;
                const int hs = (CC == 2 ? h * EB : 0);
                auto elem = [&](const uint32_t (&w)[NW], int r) -> T {
                    const int bit = r * CC * EB;                // static part of the element's bit offset
                    return wave_elem<CODE, RAW, EB>(w[bit >> 5], (bit & 31) + hs);
                };
                // quad q lives at frame byte q QB; lane l takes q = l + 32 j (j < 16) and q = (31 - l) + 32 (31 - j)
                const unsigned char* p0 = wbuf + raw_at(h, 0) + l * QB;
                const unsigned char* p1 = wbuf + raw_at(h, 0) + (31 - l) * QB;
                uint32_t wq[2][8][NW];
                auto quads = [&](int b) {                      // batch b: quads j = 8 b .. 8 b + 7
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int j = 8 * b + i;
                        const int qb = (j < 16 ? j : 31 - j) * 32 * QB;        // frame byte of the quad, without the lane term
                        const unsigned char* p = (j < 16 ? p0 : p1) + (raw_at(0, qb) - raw_at(0, 0));
                        uint32_t (&w)[NW] = wq[b & 1][i];
                        if constexpr (NW == 1) w[0] = *reinterpret_cast<const uint32_t*>(p);
                        else if constexpr (NW == 2) { const v2u v = *reinterpret_cast<const v2u*>(p); w[0] = v[0]; w[1] = v[1]; }
                        else { const v4u v = *reinterpret_cast<const v4u*>(p); w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3]; }
                    }
                };
                quads(0);
                FRAD_FENCE();
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < 3) quads(b + 1); else tw_load(0);  // the next batch (or the first twiddles) is in flight while this one is converted
                    FRAD_FENCE();
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int j = 8 * b + i;
                        z[j] = j < 16 ? cx<T>{elem(wq[b & 1][i], 0), elem(wq[b & 1][i], 2)} : cx<T>{elem(wq[b & 1][i], 3), elem(wq[b & 1][i], 1)};
                    }
                    FRAD_FENCE();
                }
            });
            team_sync<64>();                                   // raw bytes consumed: the buffer may be overwritten
        } else {
            // ---- wide elements: one global load per element, no staging ---------------------------
            const unsigned char* fb = pcm + frame_off(frame_of(u, h)) + (CC == 2 ? (h << LG) : 0);
            dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
                constexpr int CODE = decltype(code_tag)::value;
                constexpr bool RAW = decltype(raw_tag)::value != 0;
                constexpr int G = LG == 3 ? 8 : 16;            // element pairs per group of loads in flight
#pragma unroll
                for (int j0 = 0; j0 < 32; j0 += G) {
                    u64 raw[G][2];
#pragma unroll
                    for (int jj = 0; jj < G; ++jj) {
                        const int j = j0 + jj;
                        const int q = j < 16 ? l + 32 * j : (31 - l) + 32 * (31 - j);
                        const unsigned char* p = fb + (((long long)(4 * q) * CC) << LG);
                        constexpr int RS = CC << LG;           // bytes per sample-frame
                        raw[jj][0] = load_raw(p + (j < 16 ? 0 : 3) * RS, LG);
                        raw[jj][1] = load_raw(p + (j < 16 ? 2 : 1) * RS, LG);
                    }
#pragma unroll
                    for (int jj = 0; jj < G; ++jj)
                        z[j0 + jj] = cx<T>{cvt_pcm_c<T, CODE, RAW, true>(raw[jj][0]), cvt_pcm_c<T, CODE, RAW, true>(raw[jj][1])};
                    FRAD_FENCE();
                }
            });
            tw_load(0);
            FRAD_FENCE();
        }
        FRAD_STAMP(1); FRAD_MARK("pass1");
        // ---- pass 1: DFT over j --------------------------------------------------------------------
        cx<T> e[16], o[16];
        dif32_stage<false>(z, e, o);
        dft<16, false>(e);                                     // B[2 i]     (row k2 = 2 i of the exchange)
        dft<16, false>(o);                                     // B[2 i + 1]
        FRAD_FENCE();
        FRAD_STAMP(2); FRAD_MARK("twiddle_exchange");
        // ---- twiddle W_1024^(l k2) a batch of rows at a time; real parts go straight to the exchange plane -----
        T* pl = reinterpret_cast<T*>(wbuf) + h * kWavePlaneSlots;
        T* plw = pl + 2 * (l & 15) + (l >> 4);                 // this lane's column (writer side)
#pragma unroll
        for (int b = 0; b < 32 / TWB; ++b) {
            if (b < 32 / TWB - 1) tw_load(b + 1);
            FRAD_FENCE();
#pragma unroll
            for (int i = 0; i < TWB; ++i) {
                const int k2 = TWB * b + i;
                cx<T>& v = (k2 & 1) ? o[k2 >> 1] : e[k2 >> 1];
                if (k2 > 0) v = cmul(v, twb[b & 1][i]);
#ifndef FRAD_X_NOEXCH
                plw[34 * k2] = v.x;
#endif
            }
            FRAD_FENCE();
        }
        team_sync<64>();
        // ---- pass 2 inputs: even half of residue a = l (sums), odd half of residue b = -l (differences) ------
        cx<T> E[16], O[16];
        {
            const int bb = (32 - l) & 31;
            const v2d* ra = reinterpret_cast<const v2d*>(pl + 34 * l);
            const v2d* rb = reinterpret_cast<const v2d*>(pl + 34 * bb);
            constexpr int PRB = FRAD_WAVE_PRB, NQ = 16 / PRB;  // pairs per batch; batches per row
            v2d pr[2][PRB];
            auto pairs = [&](int q) {                          // batch q: the first NQ cover row a, the rest row b
#pragma unroll
                for (int i = 0; i < PRB; ++i) pr[q & 1][i] = (q < NQ ? ra : rb)[PRB * (q % NQ) + i];
            };
            auto plane = [&](auto comp) {                      // comp: 0 = real parts, 1 = imaginary parts
                constexpr int Y = decltype(comp)::value;
                pairs(0);
                FRAD_FENCE();
#pragma unroll
                for (int q = 0; q < 2 * NQ; ++q) {
                    if (q < 2 * NQ - 1) pairs(q + 1);
                    FRAD_FENCE();
#pragma unroll
                    for (int i = 0; i < PRB; ++i) {
                        const int n = PRB * (q % NQ) + i;
                        const v2d v = pr[q & 1][i];            // (y[n], y[n + 16])
                        if (q < NQ) { if constexpr (Y == 0) E[n].x = v[0] + v[1]; else E[n].y = v[0] + v[1]; }
                        else { if constexpr (Y == 0) O[n].x = v[0] - v[1]; else O[n].y = v[0] - v[1]; }
                    }
                    FRAD_FENCE();
                }
            };
#ifdef FRAD_X_NOEXCH
#pragma unroll
            for (int n = 0; n < 16; ++n) { E[n] = e[n]; O[n] = o[n]; }
#else
            plane(ic<0>{});
            team_sync<64>();
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) plw[34 * k2] = ((k2 & 1) ? o[k2 >> 1] : e[k2 >> 1]).y;
            team_sync<64>();
            plane(ic<1>{});
#endif
        }
#ifndef FRAD_HOST_EMULATION
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every plane read has returned: the DMA may overwrite the planes
#endif
        team_sync<64>();
        FRAD_STAMP(3); FRAD_MARK("pass2");
        FRAD_FENCE();
        if constexpr (MODE != 1) dma_in(next < ue ? next : ub);    // raw landing zone is free from here on (no branch; a wave's last DMA re-reads the
                                                               //  block's first unit, which its six neighbours hit in L2 -- never consumed)
                                                               //  (profile 1: issued inside the tail, whose first pass uses the whole buffer)
        FRAD_FENCE();
        // ---- pair-step tables of the first group, then pass 2 proper -----------------------------------
        // group g (processed from NG - 1 down to 0) holds jobs t and 15 - t for t in [g JPG / 2, (g + 1) JPG / 2)
        // pair-step tables (w_k, g_k of one job), fetched one job ahead of their use (double buffered)
        cx<T> ptab[2][2];
        auto job_slot = [](int g, int i) -> int { const int t = g * (JPG / 2) + (i >> 1); return (i & 1) ? 15 - t : t; };
        auto ptab_load = [&](int g, int i) {                   // job i of group g (jobs run g = NG - 1 .. 0, i = 0 .. JPG - 1)
            const int s = job_slot(g, i);
#ifdef FRAD_X_NOPTAB
            ptab[i & 1][0] = cx<T>{sc, deferred}; ptab[i & 1][1] = cx<T>{deferred, sc};
#else
            ptab[i & 1][0] = ltab[WaveLayout::PW + s * 32 + l];
            ptab[i & 1][1] = ltab[WaveLayout::PG + s * 32 + l];
#endif
        };
        dft<16, false>(E);                                     // E[u] = Z[l + 64 u]
        FRAD_FENCE();
        tw32_apply<false>(O);
        dft<16, false>(O);                                     // O[u] = Z[b + 32 (2 u + 1)]
        FRAD_FENCE();
        if constexpr (MODE == 1) {
            wave_p1_tail<CC>(E, O, ltab, smem, wbuf, payload, pw, g, u, lane, [&]() { dma_in(next < ue ? next : ub); },
                             [&](auto ph) { (void)ph; FRAD_STAMP(decltype(ph)::value); });
        } else {
        ptab_load(NG - 1, 0);
        FRAD_FENCE();
        FRAD_STAMP(4); FRAD_MARK("p0_pair_pack_store");
        // ---- DCT pair step on registers; storage codes to the staging rows, rows to HBM, group by group -------
        const bool lane0 = (l == 0);
        const long long f = CC == 2 ? u : 2 * u + h;
        const bool live = f < g.n_frames;
        unsigned char* dstf = payload + (live ? f : 0) * g.payload_stride + (CC == 2 ? lane : l) * 16;
        double fm = 0.0;
        bool nan = false;
        auto code_store = [&](unsigned char* p, T v) {         // one storage code at its payload position inside a staging row
            absmax_acc(fm, v);
            if constexpr (LG == 3) nan |= (v != v);            // only float64 PCM can carry NaN / Inf in
            if constexpr (BITS == 32) {
                *reinterpret_cast<uint32_t*>(p) = wave_perm(f2u((float)v), psel);
            } else if constexpr (BITS == 16) {
                *reinterpret_cast<unsigned short*>(p) = (unsigned short)wave_perm(f64_to_f16_bits(v), psel);
            } else {
                const u64 c = d2u(v);
                const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
                v2u w;
                if (g.le) { w[0] = lo; w[1] = hi; } else { w[0] = wave_perm(hi, psel); w[1] = wave_perm(lo, psel); }
                *reinterpret_cast<v2u*>(p) = w;
            }
        };
        // staging: buffer (g & 1) = 4 rows of 1 KiB: bin classes A = [0, 512), B = [512, 1024) (descending with g),
        // C = [1024, 1536), D = [1536, 2048) (descending).  Inside a row: payload byte order of the piece.
        int hq = h, lq = l;
        FRAD_OPAQUE(hq); FRAD_OPAQUE(lq);                     // (the staging addresses are built here, not hoisted above pass 2)
        unsigned char* stg = wbuf + (CC == 2 ? hq * NB : hq * 512);
        const int la = lq * ES, lb = (lane0 ? 32 : 64 - lq) * ES;             // bin offsets (bytes) of the two job kinds
        auto row_of = [&](int g, int cls) -> unsigned char* { return stg + (g & 1) * 4096 + cls * 1024; };
        auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
#pragma unroll
        for (int gi = NG - 1; gi >= 0; --gi) {
#pragma unroll
            for (int i = 0; i < JPG; ++i) {
                if (i + 1 < JPG) ptab_load(gi, i + 1); else if (gi > 0) ptab_load(gi - 1, 0);      // (JPG is even: the buffers keep alternating)
                FRAD_FENCE();
                const int s = job_slot(gi, i);
                const int t = s < 8 ? s : 15 - s, tl = t - gi * (JPG / 2);      // position of the job's 64 bins inside the piece
                // slot s < 8 : k = l + 64 s       , Z[k] = E[s]     , Z[M-k] = O[15-s]   (lane 0: E[(16-s) & 15])
                // slot s >= 8: k = 64 (t+1) - l   , Z[k] = O[15-s]  , Z[M-k] = E[s]      (lane 0: k = 64 t + 32, Z[M-k] = O[s])
                const cx<T> zk = s < 8 ? E[s] : O[15 - s];
                const cx<T> zm = s < 8 ? sel(E[(16 - s) & 15], O[15 - s]) : sel(O[s], E[s]);
                const cx<T> zp = conj(zm);
                const cx<T> p = cmul(zk + zp, ptab[i & 1][0]), q = cmul(zk - zp, ptab[i & 1][1]);
                const cx<T> S = p + q, D = p - q;              // (already scaled by 1/2N: the tables carry it)
                const T xm = (D.x - D.y) * K<T>::s2, xp = (D.x + D.y) * K<T>::s2;
                // byte offset of bin k inside its piece (classes A, C); classes B, D hold bin M - k / N - k at BPC - that
                const int offa = 64 * tl * ES + (s < 8 ? la : lb);
                unsigned char* pa = row_of(gi, 0) + offa;
                unsigned char* pc = row_of(gi, 2) + offa;
                unsigned char* pb = row_of(gi, 1) + BPC * ES - offa;
                unsigned char* pd = row_of(gi, 3) + BPC * ES - offa;
                if (s == 0) {
                    // lane 0 holds k = 0: X[0] (class A) and X[M] = xm, which is the first bin of class C; no X[N]
                    code_store(pa, S.x);
                    code_store(lane0 ? pa : pb, lane0 ? S.x : xm);
                    code_store(pc, lane0 ? xm : xp);
                    code_store(lane0 ? pa : pd, lane0 ? S.x : -S.y);
                } else if (tl == 0 && s < 8) {
                    // lane 0's bins M - k and N - k sit one past the piece: first bin of the next-higher piece of their
                    // class, i.e. of group gi - 1's rows (other buffer; those rows are still being filled)
                    code_store(pa, S.x);
                    code_store(lane0 ? row_of(gi - 1, 1) : pb, xm);
                    code_store(pc, xp);
                    code_store(lane0 ? row_of(gi - 1, 3) : pd, -S.y);
                } else {
                    code_store(pa, S.x);
                    code_store(pb, xm);
                    code_store(pc, xp);
                    code_store(pd, -S.y);
                }
                FRAD_FENCE();
            }
            if (gi == NG - 1) {
                // lane 0 only: the self-paired bin k = 512 gives X[512] and X[1536], the lowest bins of classes B and D: they
                // belong to the rows of this first-processed group (done after its jobs: fewer live registers by then)
                if (lane0) {
                    const cx<T> zk = E[8], zp = conj(E[8]);
                    const cx<T> p = cmul(zk + zp, ltab[WaveLayout::TW1 + 0]), q = cmul(zk - zp, ltab[WaveLayout::TW1 + 1]);
                    const cx<T> S = p + q;
                    code_store(row_of(NG - 1, 1), S.x);
                    code_store(row_of(NG - 1, 3), -S.y);
                }
            }
            FRAD_FENCE();
            team_sync<64>();
            FRAD_FENCE();
            // ---- the group's four rows: back as 16 bytes per lane, out as four 1 KiB stores -------------
            {
                v4u row[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) row[c] = *reinterpret_cast<const v4u*>(wbuf + (gi & 1) * 4096 + c * 1024 + lane * 16);
                team_sync<64>();                               // rows read: the next group may drop its stray bins into this buffer
                // frame byte offset of the piece: class A / C ascend with g, B / D descend
                const int oa = gi * 512 * CC, ob = 1024 * ES - (gi + 1) * 512 * CC, oc = 1024 * ES + gi * 512 * CC, od = 2048 * ES - (gi + 1) * 512 * CC;
                const int off[4] = {oa, ob, oc, od};
#pragma unroll
                for (int c = 0; c < 4; ++c) {                  // (exec-masked for the dead half of an odd mono tail: the instruction count stays 4)
#ifdef FRAD_WAVE_NOSTORE
                    if (live && row[c][0] == 0x12345678u) *FRAD_GPTR(v4u, dstf + off[c]) = row[c];
#else
                    if (CC == 2 || live) { if (g.fpb) *FRAD_GPTR(v4u, dstf + off[c]) = row[c]; else stream_store(dstf + off[c], row[c]); }
#endif
                }
            }
            FRAD_FENCE();
        }
        FRAD_STAMP(5);
        {
            u64 mx = nan ? 0x7ff8000000000000ULL : d2u(fm);   // np.max(np.abs(.)) propagates NaN
            if (absmax != nullptr) {
                if constexpr (CC == 2) {
                    mx = wave_max_u64(mx);
                    if (lane == 0) {
                        *FRAD_GPTR(u64, absmax + f) = mx;
                        if (g.ovf_flag != nullptr && u2d(mx) > g.ovf_limit) atomicOr(g.ovf_flag, 1);      // (NaN compares false, as in the reference)
                    }
                } else {
                    u64 m0, m1;
                    half_wave_max_u64(mx, m0, m1);
                    if (l == 0 && live) {
                        const u64 mm = h ? m1 : m0;
                        *FRAD_GPTR(u64, absmax + f) = mm;
                        if (g.ovf_flag != nullptr && u2d(mm) > g.ovf_limit) atomicOr(g.ovf_flag, 1);
                    }
                }
            }
        }
        }                                                      // MODE 0
        team_sync<64>();
        FRAD_STAMP(6); FRAD_MARK("unit_end");
#if defined(FRAD_WAVE_STAMPS) && !defined(FRAD_HOST_EMULATION)
        ++st_units;
#endif
        u = next;
    }
    FRAD_STAMP_FLUSH;
}

template <int LG, int CC, int BITS, bool CLIPS = false>
__global__ void FRAD_WAVE_BOUNDS
k_p0_fwd_wave(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<double>* __restrict__ blob, Geom g) {
    wave_fwd_body<LG, CC, BITS, 0, P1None, CLIPS>(pcm, payload, absmax, blob, g, P1None{});
}
// K7: PCM -> q int32 [n_frames, 2048, C] + pw.tq_out [n_frames, 27, C]  (full frames, integer or f64 PCM)
template <int LG, int CC>
__global__ void FRAD_WAVE_BOUNDS
k_p1_fwd_wave(const unsigned char* __restrict__ pcm, int32_t* __restrict__ q, const cx<double>* __restrict__ blob, Geom g, P1Wave pw) {
    wave_fwd_body<LG, CC, 32, 1>(pcm, reinterpret_cast<unsigned char*>(q), nullptr, blob, g, pw);
}


// =============================================================================================
// decode: payload -> float64 PCM (profile0.digital, profile0.py:46-69).  Same wave-per-unit structure, run backwards:
//   payload words of the unit's 16 pair jobs (4 bins each), fetched straight into registers during the PREVIOUS unit's
//     output phase (64 registers at 16- / 32-bit storage; issued before that unit's stores, so waiting for them never
//     waits for a store)
//   unpack + NaN/Inf scrub -> inverse DCT pair step on registers -> Z'[l + 64 u] (E) and Z'[-l + 32 (2u + 1)] (O)
//   two inverse 16-point DFTs; the odd half takes the conjugate W_32^n
//   ONE exchange: lane a writes Ee_a[n] to row a and Oo_(-a)[n] next to it in row -a; lane n2 reads the pair
//     (Ee_r, Oo_r)[n2 mod 16] of every row r with one ds_read_b128 and forms Y_r = Ee_r +- Oo_r -- the last radix-2
//     step of the 32-point inverse DFT over m, done by the reader (real plane, then imaginary plane)
//   conjugate twiddle W_1024^(r n2), inverse 32-point DFT over r -> z[n2 + 32 n1]: this channel's packed time sequence
//   output: slots n1 and 31 - n1 complete sample-frames [128 n1, 128 n1 + 128); four such blocks = 8 KiB go to an
//     XOR-swizzled staging buffer in sample order (Makhoul's permutation undone by the write addresses), come back as
//     16 bytes per lane and leave as eight coalesced 1 KiB stores -- 32 stores per unit, spread over the output phase.
// =============================================================================================
// a + s * b into a register of its own (the compiler's tied v_fmac would keep the result inside the 128-bit load tuple
// that delivered a and b, i.e. four registers alive for a two-register value)
__device__ __forceinline__ double fma_free(double s, double b, double a) {
#ifdef FRAD_HOST_EMULATION
    return fma(s, b, a);
#else
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(b), "v"(a));
    return r;
#endif
}
template <typename T> __device__ __forceinline__ cx<T> cmulc(cx<T> a, cx<T> w) {      // a * conj(w)
    return {fma(a.x, w.x, a.y * w.y), fma(a.y, w.x, -(a.x * w.y))};
}

// OUT: -1 = float64 samples out (the reference's `digital`); a FRAD_PCM_* code = the decoder's from_f64 conversion fused
// into the output stage (CC == 2): the staged float64 rows come back four sample-frames per lane, are converted and
// leave as 8 x itemsize contiguous bytes per lane -- 4-6 B per sample over HBM instead of 8.
template <int CC, int BITS, int MODE, typename P1, int OUT = -1, bool CLIPS = false>
__device__ __forceinline__ void
wave_inv_body(const unsigned char* __restrict__ payload, double* __restrict__ out, const cx<double>* __restrict__ blob, const Geom& g, const P1& pw) {
    using T = double;
    constexpr int M = 1024, N = 2048, NB = BITS / 8;
    static_assert(MODE == 0 || BITS == 32, "profile 1: the integers have the layout of a 32-bit payload");
    static_assert(BITS == 16 || BITS == 32 || BITS == 64, "whole-byte power-of-two storage");
    static_assert(CC == 1 || CC == 2, "channels");
    constexpr int FPW = 2 / CC;
    constexpr bool PF = NB <= 4;                       // the next unit's payload words are fetched across the loop edge
    using code_t = typename std::conditional<BITS == 64, u64, uint32_t>::type;
    FRAD_DYN_SMEM(smem);
    {
        cx<T>* l = reinterpret_cast<cx<T>*>(smem);
        for (int i = threadIdx.x; i < WaveLayout::SLOTS; i += blockDim.x) l[i] = blob[i];
    }
    const cx<T>* ltab = reinterpret_cast<const cx<T>*>(smem);
    if constexpr (MODE == 1) p1w_k8_tables_a(smem, pw, 1.0 / pw.scale);
#ifdef FRAD_HOST_EMULATION
    const int wv = threadIdx.x >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
    unsigned char* wbuf = smem + kWaveTableBytes + wv * kWaveBufBytes;
    const long long n_units = (g.n_frames + FPW - 1) / FPW;
    const long long ub = n_units * blockIdx.x / gridDim.x, ue = n_units * (blockIdx.x + 1) / gridDim.x;
    unsigned* ctr = reinterpret_cast<unsigned*>(smem + kWaveTableBytes + kWaveWaves * kWaveBufBytes);
    if (threadIdx.x == 0) *ctr = (unsigned)kWaveWaves;
    const uint32_t psel = g.le ? 0x07060504u : (BITS == 16 ? 0x07070405u : 0x04050607u);
    int lane = threadIdx.x & 63;

    auto frame_of = [&](long long u, int h) -> long long {
        if constexpr (CC == 2) return u;
        else { const long long f = 2 * u + h; return f < g.n_frames ? f : g.n_frames - 1; }
    };
    // payload words of unit u: w[s][c] = stored code of bin (k, N - k, M - k, M + k)[c] of job slot s, this lane's channel
    code_t w[16][4];
    auto load_words = [&](long long u, int part) {             // part p = job slots 4p .. 4p+3 (-1: all)
        const int h = lane >> 5, l = lane & 31;
        const bool lane0 = (l == 0);
        // eight lane pointers (bin classes k, M + k, M - k, N - k for the two job kinds) + the slot's constant in the
        // instruction's immediate offset
        constexpr int ES = CC * NB;
        const unsigned char* fp = payload + frame_of(u, h) * g.payload_stride + (CC == 2 ? h * NB : 0);
        const int ka = l, kb = lane0 ? 32 : 64 - l;
        const unsigned char* pA[2] = {fp + ka * ES, fp + kb * ES};                            // X[k]
        const unsigned char* pC[2] = {fp + (M + ka) * ES, fp + (M + kb) * ES};                // X[M + k]
        const unsigned char* pB[2] = {fp + (M - ka) * ES, fp + (M - kb) * ES};                // X[M - k]
        const unsigned char* pD[2] = {fp + (N - ka) * ES, fp + (N - kb) * ES};                // X[N - k]
        auto fetch = [&](const unsigned char* p) -> code_t {
#if defined(FRAD_HOST_EMULATION) || defined(FRAD_WAVE_PLAIN_LOADS)
            if constexpr (BITS == 16) return *FRAD_GCPTR(unsigned short, p);
            else if constexpr (BITS == 32) return *FRAD_GCPTR(uint32_t, p);
            else return *FRAD_GCPTR(u64, p);
#else                                                          // read once: nontemporal (keeps the streams out of each other's way in L2 / MALL)
            if (g.fpb) {
                if constexpr (BITS == 16) return *FRAD_GCPTR(unsigned short, p);
                else if constexpr (BITS == 32) return *FRAD_GCPTR(uint32_t, p);
                else return *FRAD_GCPTR(u64, p);
            }
            if constexpr (BITS == 16) return __builtin_nontemporal_load(FRAD_GCPTR(unsigned short, p));
            else if constexpr (BITS == 32) return __builtin_nontemporal_load(FRAD_GCPTR(uint32_t, p));
            else return __builtin_nontemporal_load(FRAD_GCPTR(u64, p));
#endif
        };
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (part >= 0 && (s >> 2) != part) continue;
            const int j = s < 8 ? 0 : 1, o = 64 * (s < 8 ? s : 15 - s) * ES;
            w[s][0] = fetch(pA[j] + o);
            if (s == 0) {                                      // lane 0 (k = 0): X[N] does not exist and X[M - 0] = X[M + 0]; it fetches
                w[s][1] = fetch(lane0 ? fp + 1536 * ES : pD[0]);       // the self-paired bin's X[1536] and X[512] instead (job k = 512)
                w[s][2] = fetch(lane0 ? fp + 512 * ES : pB[0]);
            } else {
                w[s][1] = fetch(pD[j] - o);
                w[s][2] = fetch(pB[j] - o);
            }
            w[s][3] = fetch(pC[j] + o);
        }
    };
    // MODE 1 (K8): the table-driven dequantiser.  Per bin: |q|^(4/3) / 2^(bits-1) from an LDS table (one multiply saved: the
    // scale is a power of two), the bin's band and position inside it from the pk map (one 8-byte read per job), the band's
    // {threshold, ramp step} as one 16-byte read, numpy's linspace arithmetic t0 + i * step (two roundings, p1tools.py:35-41),
    // one product, the sign.  What the tables do not hold -- a band code outside [0, 256), |q| >= 256 -- marks the frame for
    // the exact kernel behind this one (pw.redo); `big` collects the magnitudes.  With table thresholds (all >= 1, spaced by
    // >= 2^-52) numpy's second linspace branch (a step that underflows while the end points differ) cannot be reached.
    [[maybe_unused]] const P1K8Lds p1t = p1w_k8_lds(smem);
    [[maybe_unused]] double* const rec = reinterpret_cast<double*>(wbuf);      // MODE 1: rec[(h * 32 + band) * 2] = threshold, [+ 1] = ramp
                                                                               // step; valid from the unit's start to its pair step
    [[maybe_unused]] uint32_t big = 0;
    [[maybe_unused]] auto value_p1 = [&](uint32_t c, uint32_t en, const unsigned char* recb) -> T {
        const int32_t qv = (int32_t)c;
        const uint32_t aq = (uint32_t)(qv < 0 ? -qv : qv);
        big |= aq;
        const double dq = p1t.deqs[aq & 255u];
        const v2d rc = *reinterpret_cast<const v2d*>(recb + ((en >> 6) & 0x3f0u));
        const double y = (double)(int)(en & 1023u) * rc[1];
        const double x = dq * (y + rc[0]);
        return u2d(d2u(x) | ((u64)(c & 0x80000000u) << 32));
    };
    auto value = [&](code_t c) -> T {                          // stored code -> float64, NaN / Inf -> 0 (profile0.py:62-66)
        if constexpr (BITS == 32) {
            float f = u2f(wave_perm(c, psel));
#ifdef FRAD_HOST_EMULATION
            if (!std::isfinite(f)) f = 0.0f;
#else
            if (__builtin_amdgcn_classf(f, 0x207)) f = 0.0f;   // class mask: bit 0 signalling NaN, 1 quiet NaN, 2 -Inf, 9 +Inf
#endif
            return (T)f;
        } else if constexpr (BITS == 16) {
            float f = f16_bits_to_f32(wave_perm(c, psel) & 0xffffu);
#ifdef FRAD_HOST_EMULATION
            if (!std::isfinite(f)) f = 0.0f;
#else
            if (__builtin_amdgcn_classf(f, 0x207)) f = 0.0f;
#endif
            return (T)f;
        } else {
            const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
            const u64 v = g.le ? c : ((u64)wave_perm(lo, psel) << 32) | wave_perm(hi, psel);
            return ((v >> 52) & 0x7ff) == 0x7ff ? 0.0 : u2d(v);
        }
    };

    long long u = ub + wv;
    if constexpr (PF) { if (u < ue) load_words(u, -1); }
    // MODE 1: lane (h, l < 27)'s quantised band threshold of a unit, fetched one unit ahead like the payload words
    [[maybe_unused]] int32_t tq_pf = 0;
    [[maybe_unused]] auto tq_fetch = [&](long long un) -> int32_t {
        if constexpr (MODE == 1) {
            const int hh = (threadIdx.x >> 5) & 1, ll = threadIdx.x & 31;
            const long long ft = CC == 2 ? un : frame_of(un, hh);
            return ll < 27 ? pw.tq_in[(ft * 27 + ll) * CC + (CC == 2 ? hh : 0)] : 0;
        } else { (void)un; return 0; }
    };
    if constexpr (MODE == 1) { if (u < ue) tq_pf = tq_fetch(u); }
    __syncthreads();                                          // tables and counter are in LDS
    if constexpr (MODE == 1) { p1w_k8_tables_b(smem, pw); __syncthreads(); }
    for (int i = (wv * 8 + (int)(blockIdx.x & 7)) * g.cg; i > 0; --i) FRAD_WAVE_SLEEP(1);      // start stagger (g.cg x 64 cycles per step; 0 = off)
    while (u < ue) {
        lane = threadIdx.x & 63; FRAD_OPAQUE(lane);
        const int h = lane >> 5, l = lane & 31;
        const bool lane0 = (l == 0);
        const long long next = ub + wave_next_unit(ctr);
        if constexpr (!PF) load_words(u, -1);
        [[maybe_unused]] bool bad = false;                     // MODE 1: this lane saw something the tables do not hold
        if constexpr (MODE == 1) {
            // thresholds of this unit's frame(s): thr[b] = (e/2)^quant(tq[b]) (profile1.py:63), ramp steps (p1tools.py:35-41)
            big = 0;
            if (l < 27) {
                const int32_t ti = tq_pf;
                bad = (uint32_t)ti >= 256u;
                rec[(h * 32 + l) * 2] = p1t.thrt[bad ? 0 : ti];
            } else if (l == 31) {
                rec[(h * 32 + 31) * 2] = 0.0; rec[(h * 32 + 31) * 2 + 1] = 0.0;       // bins past the last band start (pk band 31)
            }
            team_sync<64>();
            if (l < 27) {
                double st = 0.0;
                if (l < 26) { const int num = p1t.edge[l + 1] - p1t.edge[l]; if (num > 0) st = (rec[(h * 32 + l + 1) * 2] - rec[(h * 32 + l) * 2]) / (double)num; }
                rec[(h * 32 + l) * 2 + 1] = st;
            }
            team_sync<64>();
        }
        // ---- inverse pair step: X[k], X[N-k], X[M-k], X[M+k] -> Z'[k], Z'[M-k] -----------------------
        cx<T> E[16], O[16];
        {
            cx<T> zk[16], zm[16];
            cx<T> ptab[2][2];
            [[maybe_unused]] v2u ent[2];                       // MODE 1: the job's four (band, position) entries, fetched a job ahead
            [[maybe_unused]] const unsigned char* recb = wbuf + h * 512;
            auto ptab_load = [&](int s) {
                ptab[s & 1][0] = ltab[WaveLayout::PW + s * 32 + l];
                ptab[s & 1][1] = ltab[WaveLayout::PG + s * 32 + l];
                if constexpr (MODE == 1) ent[s & 1] = reinterpret_cast<const v2u*>(p1t.pk)[32 * s + l];
            };
            auto inv_pair = [&](T xk, T xnk, T a, T b, cx<T> wk, cx<T> gk, cx<T>& rk, cx<T>& rm) {
                const cx<T> uu = {xk, -xnk};
                const cx<T> sv = {(a + b) * K<T>::s2, (b - a) * K<T>::s2};
                const cx<T> A = cmulc(uu + sv, wk), B = cmulc(uu - sv, gk);
                rk = A + B; rm = conj(A - B);
            };
            ptab_load(0);
            FRAD_FENCE();
            cx<T> zsp, dump;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) ptab_load(s + 1);
                FRAD_FENCE();
                // bins of the four words (see load_words): k, N - k, M - k, M + k; lane 0 of job 0 borrows 1536 and 512
                T xa, xd, xb, xc;
                if constexpr (MODE == 1) {
                    // pk classes (k7_bin_of): 0 = k, 1 = M - k, 2 = M + k, 3 = N - k; lane 0 of job 0: 0, 512, 1024, 1536
                    const v2u e2 = ent[s & 1];
                    xa = value_p1(w[s][0], e2[0] & 0xffffu, recb); xd = value_p1(w[s][1], e2[1] >> 16, recb);
                    xb = value_p1(w[s][2], e2[0] >> 16, recb); xc = value_p1(w[s][3], e2[1] & 0xffffu, recb);
                } else {
                    xa = value(w[s][0]); xd = value(w[s][1]); xb = value(w[s][2]); xc = value(w[s][3]);
                }
                if (s == 0) {
                    // lane 0: k = 0 (X[N] = 0, X[M] on both sides) and, from the two borrowed words, the self-paired k = 512
                    inv_pair(xa, lane0 ? 0.0 : xd, lane0 ? xc : xb, xc, ptab[0][0], ptab[0][1], zk[0], zm[0]);
                    inv_pair(xb, xd, xb, xd, ltab[WaveLayout::TW1 + 0], ltab[WaveLayout::TW1 + 1], zsp, dump);
                } else {
                    inv_pair(xa, xd, xb, xc, ptab[s & 1][0], ptab[s & 1][1], zk[s], zm[s]);
                }
                if constexpr (MODE == 1) {                     // the job's arithmetic stays in the job (the scheduling fence alone does not
                    FRAD_PIN(zk[s].x); FRAD_PIN(zk[s].y); FRAD_PIN(zm[s].x); FRAD_PIN(zm[s].y);   //  hold pure arithmetic)
                }
                FRAD_FENCE();
            }
            auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
#pragma unroll
            for (int j = 0; j < 8; ++j) { E[j] = zk[j]; O[j] = zk[15 - j]; }
#pragma unroll
            for (int j = 8; j < 16; ++j) {
                O[j] = sel(zm[j], zm[15 - j]);
                E[j] = sel(j == 8 ? zsp : zm[16 - j], zm[j]);
            }
        }
        FRAD_FENCE();
        if constexpr (MODE == 1) {
            team_sync<64>();                                   // thresholds read by every lane before the planes overwrite them
            bad |= big > 255u;
            if (wave_any(bad)) {                               // rare: hand the frame(s) to the exact kernel
                if constexpr (CC == 2) {
                    if (lane == 0) { const int i = atomicAdd(pw.redo, 1); pw.redo[1 + i] = (int)u; }
                } else {
                    const bool lo = wave_any(bad && h == 0), hi = wave_any(bad && h == 1);
                    const long long ff = 2 * u + h;
                    if (l == 0 && (h ? hi : lo) && ff < g.n_frames) { const int i = atomicAdd(pw.redo, 1); pw.redo[1 + i] = (int)ff; }
                }
            }
        }
        // ---- inverse 16-point DFTs over m; the odd half takes conj(W_32^n) --------------------------
        dft<16, true>(E);
        FRAD_FENCE();
        dft<16, true>(O);
        tw32_apply<true>(O);
        FRAD_FENCE();
        // ---- exchange: Ee to row a, Oo to row -a (next to the Ee of that row); reader forms Ee +- Oo ----
        T* pl = reinterpret_cast<T*>(wbuf) + h * kWavePlaneSlots;
        cx<T> Y[32];
        {
            const int bb = (32 - l) & 31;
            T* wa = pl + 34 * l;
            T* wb = pl + 34 * bb + 1;
            const v2d* rd = reinterpret_cast<const v2d*>(pl + 2 * (l & 15));
            const T sg = (l & 16) ? -1.0 : 1.0;
            constexpr int PRB = FRAD_WAVE_PRB;
            v2d pr[2][PRB];
            auto pairs = [&](int q) {
#pragma unroll
                for (int i = 0; i < PRB; ++i) pr[q & 1][i] = rd[17 * (PRB * q + i)];      // row r = PRB q + i: 34 doubles = 17 pairs apart
            };
            auto plane = [&](auto comp) {
                constexpr int Yc = decltype(comp)::value;
#pragma unroll
                for (int n = 0; n < 16; ++n) { wa[2 * n] = Yc ? E[n].y : E[n].x; wb[2 * n] = Yc ? O[n].y : O[n].x; }
                team_sync<64>();
                pairs(0);
                FRAD_FENCE();
#pragma unroll
                for (int q = 0; q < 32 / PRB; ++q) {
                    if (q + 1 < 32 / PRB) pairs(q + 1);
                    FRAD_FENCE();
#pragma unroll
                    for (int i = 0; i < PRB; ++i) {
                        const v2d v = pr[q & 1][i];
                        if constexpr (Yc == 0) Y[PRB * q + i].x = fma_free(sg, v[1], v[0]); else Y[PRB * q + i].y = fma_free(sg, v[1], v[0]);
                    }
                    FRAD_FENCE();
                }
            };
            plane(ic<0>{});
            team_sync<64>();
            plane(ic<1>{});
            team_sync<64>();
        }
        // ---- conjugate twiddle W_1024^(r n2), inverse 32-point DFT over r -> z[n2 + 32 n1] ------------------
        cx<T> e[16], o[16];
        {
            const cx<T>* tw = ltab + WaveLayout::TW1 + l;
            constexpr int TWB = FRAD_WAVE_TWB;
            cx<T> twb[2][TWB];
            auto tw_load = [&](int b) {
#pragma unroll
                for (int i = 0; i < TWB; ++i) if (TWB * b + i > 0) twb[b & 1][i] = tw[(TWB * b + i) * 32];
            };
            tw_load(0);
            FRAD_FENCE();
#pragma unroll
            for (int b = 0; b < 32 / TWB; ++b) {
                if (b + 1 < 32 / TWB) tw_load(b + 1);
                FRAD_FENCE();
#pragma unroll
                for (int i = 0; i < TWB; ++i) { const int r = TWB * b + i; if (r > 0) Y[r] = cmulc(Y[r], twb[b & 1][i]); }
                FRAD_FENCE();
            }
            dif32_stage<true>(Y, e, o);
            dft<16, true>(e);                                  // z[n2 + 32 (2 i)]
            FRAD_FENCE();
            dft<16, true>(o);                                  // z[n2 + 32 (2 i + 1)]
            FRAD_FENCE();
        }
        // ---- output: four groups of four 128-sample-frame blocks through the swizzled staging buffer ---------
        {
            const long long f = CC == 2 ? u : 2 * u + h;
            const bool live = f < g.n_frames;
            int lq = l, hq = h;
            FRAD_OPAQUE(lq); FRAD_OPAQUE(hq);
            long long fbase;                                      // first sample-frame of the output frame
            if constexpr (CLIPS) { const long long ff = live ? f : 0; fbase = ff * (long long)N + (long long)((unsigned)ff / (unsigned)g.fpc) * (g.clip_stride - (long long)g.fpc * N); }
            else fbase = (live ? f : 0) * (long long)N;
            unsigned char* dstf = reinterpret_cast<unsigned char*>(out + fbase * CC);
            // staging position of local sample-frame S = 4 n + r (n = lane's quad inside the block, r = 0..3), see header:
            //   CC == 2: 16-byte rows R = S, physical row R ^ ((R >> 3) & 7), channel h in the row's half
            //   CC == 1: 8-byte slots S of frame h (4 KiB per frame and group), physical slot S ^ (((S >> 4) & 3) << 1)
            auto spos = [&](int n, int r) -> int {             // byte offset inside a group buffer, without the block term
                const int S = 4 * n + r;
                if constexpr (CC == 2) return ((S ^ ((S >> 3) & 7)) * 16) + 8 * hq;
                else return hq * 4096 + ((S ^ (((S >> 4) & 3) << 1)) * 8);
            };
            const int pa0 = spos(lq, 0), pa2 = spos(lq, 2), pb3 = spos(31 - lq, 3), pb1 = spos(31 - lq, 1);
            const int rdo = CC == 2 ? ((lane ^ ((lane >> 3) & 7)) * 16) : (hq * 4096 + ((lq ^ ((lq >> 3) & 3)) * 16));
            constexpr int BLK = CC == 2 ? 2048 : 1024;         // bytes of one 128-sample-frame block (per frame)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                unsigned char* sb = wbuf + (gq & 1) * 8192;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n1 = 4 * gq + j, i = n1 >> 1;
                    const cx<T> zlo = (n1 & 1) ? o[i] : e[i], zhi = (n1 & 1) ? e[15 - i] : o[15 - i];
                    unsigned char* bp = sb + j * BLK;
                    *reinterpret_cast<T*>(bp + pa0) = zlo.x;   // x[4 q]
                    *reinterpret_cast<T*>(bp + pa2) = zlo.y;   // x[4 q + 2]
                    *reinterpret_cast<T*>(bp + pb3) = zhi.x;   // x[4 q' + 3]
                    *reinterpret_cast<T*>(bp + pb1) = zhi.y;   // x[4 q' + 1]
                }
                FRAD_FENCE();
                // a quarter of the next unit's payload words per group, into the registers this group's samples have just
                // freed (the first quarter before any store of this unit: waiting for it never waits for a store)
                if constexpr (PF) { load_words(next < ue ? next : u, gq); }
                if constexpr (MODE == 1) { if (gq == 0) tq_pf = tq_fetch(next < ue ? next : u); }
                team_sync<64>();
                FRAD_FENCE();
                if constexpr (OUT >= 0 && CC == 1) {
                    // narrowed output, mono: lane (h, t) takes samples 256 hf + 8 t .. + 7 of its frame's 512 in the group (four
                    // staged 16-byte pairs; the slot swizzle permutes pairs inside a 16-slot run), converts them and stores
                    // 8 x itemsize contiguous bytes
                    constexpr int okind = OUT >> 3, olg = (OUT >> 1) & 3, osz = 1 << olg;
                    unsigned char* dn = reinterpret_cast<unsigned char*>(out) + (fbase + 512 * gq) * osz;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        v4u r4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int G = 256 * hf + 8 * lq + 2 * i, jb = G >> 7, S = G & 127;
                            r4[i] = *reinterpret_cast<const v4u*>(sb + jb * BLK + hq * 4096 + ((S ^ (((S >> 4) & 3) << 1)) * 8));
                        }
                        if (hf == 1) team_sync<64>();
                        u64 b[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            b[2 * i] = from_f64_bits<okind, olg>(u2d((u64)r4[i][0] | ((u64)r4[i][1] << 32)), false);
                            b[2 * i + 1] = from_f64_bits<okind, olg>(u2d((u64)r4[i][2] | ((u64)r4[i][3] << 32)), false);
                        }
                        unsigned char* dl = dn + (256 * hf + 8 * lq) * osz;
                        if constexpr (olg == 1) {
                            const v4u o4 = {(uint32_t)(b[0] | (b[1] << 16)), (uint32_t)(b[2] | (b[3] << 16)), (uint32_t)(b[4] | (b[5] << 16)), (uint32_t)(b[6] | (b[7] << 16))};
                            if (live) stream_store(dl, o4);
                        } else if constexpr (olg == 2) {
                            const v4u o4 = {(uint32_t)b[0], (uint32_t)b[1], (uint32_t)b[2], (uint32_t)b[3]}, o5 = {(uint32_t)b[4], (uint32_t)b[5], (uint32_t)b[6], (uint32_t)b[7]};
                            if (live) { stream_store(dl, o4); stream_store(dl + 16, o5); }
                        } else {
                            static_assert(olg == 1 || olg == 2, "fused output conversion: 2- and 4-byte formats");
                        }
                        FRAD_FENCE();
                    }
                    FRAD_FENCE();
                    continue;
                } else
                if constexpr (OUT >= 0) {
                    // narrowed output: lane t of half hf takes sample-frames 256 hf + 4 t .. + 3 of the group (four staged rows),
                    // converts the eight values and stores 8 x itemsize contiguous bytes
                    constexpr int okind = OUT >> 3, olg = (OUT >> 1) & 3, osz = 1 << olg;
                    unsigned char* dn = reinterpret_cast<unsigned char*>(out) + (fbase + 512 * gq) * (2 * osz);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        v4u r4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int G = 256 * hf + 4 * lane + i, jb = G >> 7, R = G & 127;
                            r4[i] = *reinterpret_cast<const v4u*>(sb + jb * BLK + ((R ^ ((R >> 3) & 7)) * 16));
                        }
                        if (hf == 1) team_sync<64>();
                        u64 b[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            b[2 * i] = from_f64_bits<okind, olg>(u2d((u64)r4[i][0] | ((u64)r4[i][1] << 32)), false);
                            b[2 * i + 1] = from_f64_bits<okind, olg>(u2d((u64)r4[i][2] | ((u64)r4[i][3] << 32)), false);
                        }
                        unsigned char* dl = dn + (256 * hf + 4 * lane) * (2 * osz);
                        if constexpr (olg == 1) {
                            const v4u o4 = {(uint32_t)(b[0] | (b[1] << 16)), (uint32_t)(b[2] | (b[3] << 16)), (uint32_t)(b[4] | (b[5] << 16)), (uint32_t)(b[6] | (b[7] << 16))};
                            stream_store(dl, o4);
                        } else if constexpr (olg == 2) {
                            const v4u o4 = {(uint32_t)b[0], (uint32_t)b[1], (uint32_t)b[2], (uint32_t)b[3]}, o5 = {(uint32_t)b[4], (uint32_t)b[5], (uint32_t)b[6], (uint32_t)b[7]};
                            stream_store(dl, o4); stream_store(dl + 16, o5);
                        } else {
                            static_assert(olg == 1 || olg == 2, "fused output conversion: 2- and 4-byte formats");
                        }
                        FRAD_FENCE();
                    }
                    FRAD_FENCE();
                    continue;
                }
                // CC == 2: the group's 512 rows are 8 KiB of the frame, in order.  CC == 1: 256 rows (4 KiB) per frame.
                unsigned char* dp = dstf + gq * (CC == 2 ? 8192 : 4096) + (CC == 2 ? lane : l) * 16;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {               // four rows at a time: 16 registers
                    v4u row[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) row[i] = *reinterpret_cast<const v4u*>(sb + rdo + (4 * hf + i) * (CC == 2 ? 1024 : 512));
                    if (hf == 1) team_sync<64>();              // all rows read: the buffer may be refilled (two groups from now)
#pragma unroll
                    for (int i = 0; i < 4; ++i)            // (CC == 2: every lane is live -- no branch, so that the compiler can count these stores
                        if (CC == 2 || live) stream_store(dp + (4 * hf + i) * (CC == 2 ? 1024 : 512), row[i]);     //  when it waits for the loads issued before them)
                    FRAD_FENCE();
                }
                FRAD_FENCE();
            }
        }
        team_sync<64>();
        u = next;
    }
}

template <int CC, int BITS, bool CLIPS = false>
__global__ void FRAD_WAVE_BOUNDS
k_p0_inv_wave(const unsigned char* __restrict__ payload, double* __restrict__ out, const cx<double>* __restrict__ blob, Geom g) {
    wave_inv_body<CC, BITS, 0, P1None, -1, CLIPS>(payload, out, blob, g, P1None{});
}
// profile 0 decode straight to a narrower PCM format (stereo frames, or pairs of mono frames): frad_p0_digital_pcm
template <int BITS, int OUT, int CC = 2>
__global__ void FRAD_WAVE_BOUNDS
k_p0_inv_wave_pcm(const unsigned char* __restrict__ payload, void* __restrict__ out, const cx<double>* __restrict__ blob, Geom g) {
    wave_inv_body<CC, BITS, 0, P1None, OUT>(payload, static_cast<double*>(out), blob, g, P1None{});
}
// K8: q int32 [n_frames, 2048, C] (as `payload`, stride 2048 * C * 4 bytes) + pw.tq_in -> float64 PCM
template <int CC>
__global__ void FRAD_WAVE_BOUNDS
k_p1_inv_wave(const unsigned char* __restrict__ q, double* __restrict__ out, const cx<double>* __restrict__ blob, Geom g, P1Wave pw) {
    wave_inv_body<CC, 32, 1>(q, out, blob, g, pw);
}

}  // namespace frad
