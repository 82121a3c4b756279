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
//   ONE exchange through LDS (XOR-swizzled, conflict-free both ways)
//   pass 2: lane a needs Z[a + 32 m] AND its DCT partners Z[M - k], which live in residue -a.  It therefore
//           computes the even-m half of residue a and the odd-m half of residue -a (a 32-point DFT splits into
//           two 16-point DFTs of the sums and the twiddled differences: half the outputs for half the work), so
//           both members of every (k, M-k) pair meet in one lane and the DCT pair step runs on registers.  The
//           price is reading the exchange buffer twice (LDS reads are 3x cheaper than writes on CDNA4).
//   pack: storage codes go to LDS at their payload position, come back as 16-byte rows, coalesced stores.
//
// No s_barrier, no inter-wave traffic: the LDS serves one wave's instructions in order, so phases only need the
// compiler kept honest (team_sync<64>).  One wave per SIMD (4 x 32 KiB + 32 KiB of tables = 160 KiB), up to 512
// VGPRs each; the next frame's PCM is prefetched into registers during the transform.
#pragma once
#include "frad_kernels.hpp"

namespace frad {

// LDS table blob (complex<double> slots), shared by the four waves of a block:
//   TW1[k2 * 32 + l] = W_1024^(l * k2)                (row 0 is never used as a twiddle: slots 0, 1 hold w_512, g_512)
//   PW [u * 32 + a]  = w_k, PG[u * 32 + a] = g_k     for the pair job of (lane a, slot u), k = wave_job_k(a, u)
struct WaveLayout { static constexpr int TW1 = 0, PW = 1024, PG = 1536, SLOTS = 2048; };
constexpr int kWaveTableBytes = WaveLayout::SLOTS * 16;
constexpr int kWaveBufBytes = 32768;
constexpr int kWaveLdsBytes = kWaveTableBytes + 4 * kWaveBufBytes;     // = 160 KiB

// pair job (lane a in [0, 32), slot u in [0, 16)) -> its k in [0, M/2]; see frad_wave.hpp header and pass 2 below
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
__device__ __forceinline__ void dif32_stage(const cx<T> (&lo)[16], const cx<T> (&hi)[16], cx<T> (&e)[16], cx<T> (&o)[16]) {
    if constexpr (I < 16) {
        e[I] = lo[I] + hi[I];
        o[I] = mul_w32<I, INV>(lo[I] - hi[I]);
        dif32_stage<INV, T, I + 1>(lo, hi, e, o);
    }
}
template <bool INV, typename T, int I = 0>
__device__ __forceinline__ void tw32_apply(cx<T> (&o)[16]) {          // o[n] *= W_32^n
    if constexpr (I < 16) { o[I] = mul_w32<I, INV>(o[I]); tw32_apply<INV, T, I + 1>(o); }
}

// XOR-swizzled exchange slot of (row r, column c), r and c in [0, 32): 32 r + (c ^ (r & 15)).  Writers hold c = lane
// and sweep r (8 consecutive lanes -> 8 distinct 16-byte columns), readers hold r = lane and sweep c (16 lanes of a
// ds_read_b128 group -> 16 distinct columns): no bank conflicts either way.
__device__ __forceinline__ int xslot(int r, int c) { return 32 * r + (c ^ (r & 15)); }

#ifndef FRAD_HOST_EMULATION
#define FRAD_WAVE_BOUNDS __launch_bounds__(256, 1)
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

// =============================================================================================
// encode: PCM -> payload.  grid = min(ceil(units / 4), CUs), block = 256 (4 independent waves).
// unit = one frame (CC == 2) or a pair of frames 2u, 2u+1 (CC == 1).
// Requires: pcm 16-byte aligned, frame byte stride % 16 == 0, payload 16-byte aligned, payload_stride % 16 == 0.
// =============================================================================================
template <int LG, int CC, int BITS>
__global__ void FRAD_WAVE_BOUNDS
k_p0_fwd_wave(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<double>* __restrict__ blob, Geom g) {
    using T = double;
    constexpr int M = 1024, N = 2048, ISZ = 1 << LG, NB = BITS / 8;
    static_assert(BITS == 16 || BITS == 32 || BITS == 64, "whole-byte power-of-two storage");
    static_assert(CC == 1 || CC == 2, "channels");
    constexpr int RAWB = N * ISZ * CC;                 // raw bytes of one frame
    constexpr int FPW = 2 / CC;                        // frames per wave
    constexpr int NPF = RAWB * FPW / 1024;             // 16-byte chunks per lane and unit (= 4 ISZ)
    constexpr int PAYB = N * CC * NB;                  // payload bytes of one frame
    constexpr int NST = PAYB * FPW / 1024;             // 16-byte store chunks per lane (= 4 NB)
    constexpr int QB = 4 * CC * ISZ;                   // bytes of four consecutive sample-frames
    static_assert(RAWB * FPW <= kWaveBufBytes && PAYB * FPW <= kWaveBufBytes, "one unit fits the wave's LDS buffer");
    FRAD_DYN_SMEM(smem);
    {
        cx<T>* l = reinterpret_cast<cx<T>*>(smem);
        for (int i = threadIdx.x; i < WaveLayout::SLOTS; i += blockDim.x) l[i] = blob[i];
    }
    const cx<T>* ltab = reinterpret_cast<const cx<T>*>(smem);
    const int wv = threadIdx.x >> 6;
    unsigned char* wbuf = smem + kWaveTableBytes + wv * kWaveBufBytes;
    const long long frameb = (g.frame_stride * CC) << LG;
    const long long n_units = (g.n_frames + FPW - 1) / FPW;
    const long long stride = (long long)gridDim.x * 4;
    const bool le = g.le != 0;
    uint32_t pf[NPF][4];
    int lane = threadIdx.x & 63;

    auto frame_of = [&](long long u, int h) -> long long {   // frame this lane works on (clamped for the odd mono tail)
        if constexpr (CC == 2) return u;
        else { const long long f = 2 * u + h; return f < g.n_frames ? f : g.n_frames - 1; }
    };
    auto prefetch = [&](long long u) {
        const int h = lane >> 5, l = lane & 31;
        const unsigned char* src = pcm + frame_of(u, h) * frameb + (CC == 2 ? lane : l) * 16;
#pragma unroll
        for (int i = 0; i < NPF; ++i) load_words<4>(src + (long long)i * (CC == 2 ? 1024 : 512), pf[i]);
    };

    const T deferred = (T)pcm_deferred_scale(g.dtype, g.raw_be);
    const T sc = ((T)1 / (T)(2 * N)) * deferred, sc2 = (K<T>::s2 / (T)(2 * N)) * deferred;
    long long u = (long long)blockIdx.x * 4 + wv;
    if (u < n_units) prefetch(u);
    __syncthreads();                                          // tables are in LDS
    while (u < n_units) {
        lane = threadIdx.x & 63; FRAD_OPAQUE(lane);           // per-lane addresses are rebuilt each unit (no LICM register hoard)
        const int h = lane >> 5, l = lane & 31;
        // ---- stage the raw bytes --------------------------------------------------------------
        {
            unsigned char* dst = wbuf + (CC == 2 ? lane * 16 : h * RAWB + l * 16);
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                v4u v = {pf[i][0], pf[i][1], pf[i][2], pf[i][3]};
                *reinterpret_cast<v4u*>(dst + i * (CC == 2 ? 1024 : 512)) = v;
            }
        }
        const long long next = u + stride;
        if (next < n_units) prefetch(next);                   // lands during the transform
        team_sync<64>();
        // ---- Makhoul pairs -> z[j] = z_packed[l + 32 j] ------------------------------------------
        cx<T> z[32];
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            const unsigned char* fb = wbuf + (CC == 1 ? h * RAWB : 0);
            if constexpr (QB <= 16) {
                // whole quads: lanes read consecutive QB-byte groups (conflict-free), elements are picked by shifts
                constexpr int NW = QB / 4;
                constexpr int EB = 8 * ISZ;                     // bits per element
                const int hs = (CC == 2 ? h * EB : 0);
                auto elem = [&](const uint32_t (&w)[NW], int r) -> T {
                    const int bit = r * CC * EB;                // static part
                    uint32_t raw;
                    if constexpr (EB == 32) raw = w[bit >> 5];
                    else raw = (w[bit >> 5] >> ((bit & 31) + hs)) & ((1u << EB) - 1u);
                    return cvt_pcm_c<T, CODE, RAW, true>((u64)raw);
                };
                const unsigned char* p0 = fb + l * QB;
                const unsigned char* p1 = fb + (31 - l) * QB;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    uint32_t w[NW];
                    const unsigned char* p = j < 16 ? p0 + j * 32 * QB : p1 + (31 - j) * 32 * QB;
                    if constexpr (NW == 1) w[0] = *reinterpret_cast<const uint32_t*>(p);
                    else if constexpr (NW == 2) { const v2u v = *reinterpret_cast<const v2u*>(p); w[0] = v[0]; w[1] = v[1]; }
                    else { const v4u v = *reinterpret_cast<const v4u*>(p); w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3]; }
                    z[j] = j < 16 ? cx<T>{elem(w, 0), elem(w, 2)} : cx<T>{elem(w, 3), elem(w, 1)};
                }
            } else {
                // wide elements (4 or 8 bytes, QB = 32 / 64): one LDS read per element
                auto elem = [&](int n) -> T {
                    const unsigned char* p = fb + ((long long)(n * CC + (CC == 2 ? h : 0)) << LG);
                    u64 raw;
                    if constexpr (LG == 2) raw = *reinterpret_cast<const uint32_t*>(p);
                    else { const v2u v = *reinterpret_cast<const v2u*>(p); raw = (u64)v[0] | ((u64)v[1] << 32); }
                    return cvt_pcm_c<T, CODE, RAW, true>(raw);
                };
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int q = j < 16 ? l + 32 * j : (31 - l) + 32 * (31 - j);
                    z[j] = j < 16 ? cx<T>{elem(4 * q), elem(4 * q + 2)} : cx<T>{elem(4 * q + 3), elem(4 * q + 1)};
                }
            }
        });
        team_sync<64>();                                       // raw bytes consumed: the buffer may be overwritten
        // ---- pass 1: DFT over j, twiddle, exchange ---------------------------------------------
        cx<T>* xb = reinterpret_cast<cx<T>*>(wbuf) + h * M;
        {
            cx<T> lo[16], hi[16], e[16], o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { lo[i] = z[i]; hi[i] = z[i + 16]; }
            dif32_stage<false>(lo, hi, e, o);
            dft<16, false>(e);                                 // B[2 i]
            dft<16, false>(o);                                 // B[2 i + 1]
            const cx<T>* tw = ltab + WaveLayout::TW1 + l;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const cx<T> te = i == 0 ? e[0] : cmul(e[i], tw[(2 * i) * 32]);
                const cx<T> to = cmul(o[i], tw[(2 * i + 1) * 32]);
                xb[xslot(2 * i, l)] = te;
                xb[xslot(2 * i + 1, l)] = to;
            }
        }
        team_sync<64>();
        // ---- pass 2: even half of residue a = l, odd half of residue b = -l ----------------------
        cx<T> E[16], O[16];
        {
            const int b = (32 - l) & 31;
            cx<T> lo[16], hi[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) { lo[n] = xb[xslot(l, n)]; hi[n] = xb[xslot(l, n + 16)]; }
#pragma unroll
            for (int n = 0; n < 16; ++n) E[n] = lo[n] + hi[n];
#pragma unroll
            for (int n = 0; n < 16; ++n) { lo[n] = xb[xslot(b, n)]; hi[n] = xb[xslot(b, n + 16)]; }
#pragma unroll
            for (int n = 0; n < 16; ++n) O[n] = lo[n] - hi[n];
            tw32_apply<false>(O);
            dft<16, false>(E);                                 // E[u] = Z[l + 64 u]
            dft<16, false>(O);                                 // O[u] = Z[b + 32 (2 u + 1)]
        }
        team_sync<64>();                                       // exchange consumed: the buffer takes the payload image
        // ---- DCT pair step on registers, storage codes to their payload position ---------------
        const bool lane0 = (l == 0);
        unsigned char* img = wbuf + (CC == 1 ? h * PAYB : 0);
        const int cofs = (CC == 2 ? h : 0);
        double fm = 0.0;
        bool nan = false;
        auto put = [&](int k, T v) {                          // X[k] of this lane's channel
            fm = fmax(fm, fabs(v));
            nan |= (v != v);
            unsigned char* p = img + (k * CC + cofs) * NB;
            if constexpr (BITS == 32) {
                const uint32_t c = f2u((float)v);
                *reinterpret_cast<uint32_t*>(p) = le ? c : bswap32(c);
            } else if constexpr (BITS == 16) {
                const uint32_t c = f64_to_f16_bits(v);
                *reinterpret_cast<unsigned short*>(p) = (unsigned short)(le ? c : bswap16(c));
            } else {
                const u64 c = le ? d2u(v) : bswap64(d2u(v));
                v2u w = {(uint32_t)c, (uint32_t)(c >> 32)};
                *reinterpret_cast<v2u*>(p) = w;
            }
        };
        auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
        // lane 0 only: the self-paired bin k = 512 (its values replace the out-of-range / duplicate outputs of k = 0)
        cx<T> S512;
        {
            const cx<T> zk = E[8], zp = conj(E[8]);
            const cx<T> p = cmul(zk + zp, ltab[WaveLayout::TW1 + 0]), q = cmul(zk - zp, ltab[WaveLayout::TW1 + 1]);
            S512 = p + q;
        }
        const int klo = l, khi = lane0 ? 992 : 1024 - l;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // slot s < 8 : k = l + 64 s       , Z[k] = E[s]     , Z[M-k] = O[15-s]   (lane 0: E[(16-s) & 15])
            // slot s >= 8: k = M - l - 64 s   , Z[k] = O[15-s]  , Z[M-k] = E[s]      (lane 0: k = 992 - 64 s, Z[M-k] = O[s])
            const int k = s < 8 ? klo + 64 * s : khi - 64 * s;
            const cx<T> zk = s < 8 ? E[s] : O[15 - s];
            const cx<T> zm = s < 8 ? sel(E[(16 - s) & 15], O[15 - s]) : sel(O[s], E[s]);
            const cx<T> zp = conj(zm);
            const cx<T> p = cmul(zk + zp, ltab[WaveLayout::PW + s * 32 + l]), q = cmul(zk - zp, ltab[WaveLayout::PG + s * 32 + l]);
            const cx<T> S = p + q, D = p - q;
            if (s == 0) {
                put(k, S.x * sc);
                put(lane0 ? 1536 : N - k, (lane0 ? -S512.y : -S.y) * sc);
                put(M - k, (D.x - D.y) * sc2);
                put(lane0 ? 512 : M + k, lane0 ? S512.x * sc : (D.x + D.y) * sc2);
            } else {
                put(k, S.x * sc);
                put(N - k, -S.y * sc);
                put(M - k, (D.x - D.y) * sc2);
                put(M + k, (D.x + D.y) * sc2);
            }
        }
        team_sync<64>();
        // ---- payload rows out ---------------------------------------------------------------------
        {
            const long long f = CC == 2 ? u : 2 * u + h;
            const bool live = f < g.n_frames;
            const unsigned char* src = wbuf + (CC == 2 ? lane * 16 : h * PAYB + l * 16);
            unsigned char* dst = payload + (live ? f : 0) * g.payload_stride + (CC == 2 ? lane : l) * 16;
            if (live) {
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    const v4u v = *reinterpret_cast<const v4u*>(src + i * (CC == 2 ? 1024 : 512));
                    *FRAD_GPTR(v4u, dst + i * (CC == 2 ? 1024 : 512)) = v;
                }
            }
            u64 mx = nan ? 0x7ff8000000000000ULL : d2u(fm);   // np.max(np.abs(.)) propagates NaN
            if (absmax != nullptr) {
                if constexpr (CC == 2) {
                    mx = wave_max_u64(mx);
                    if (lane == 0) *FRAD_GPTR(u64, absmax + f) = mx;
                } else {
                    u64 m0, m1;
                    half_wave_max_u64(mx, m0, m1);
                    if (l == 0 && live) *FRAD_GPTR(u64, absmax + f) = h ? m1 : m0;
                }
            }
        }
        team_sync<64>();                                       // image read: the buffer may take the next unit's raw bytes
        u = next;
    }
}

}  // namespace frad
