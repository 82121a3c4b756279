// frad_common.hpp -- element conversion and bit-depth pack/unpack for gfx950 (CDNA4).
// Device-side building blocks shared by every kernel of libfrad_hip.so.
//
// Reference semantics restated here (paths relative to /root/reference/src/libfrad/):
//   to_f64                 backend/pcmformat.py:34-47
//   cast + truncate (pack) fourier/profile0.py:28-42   (== profile4.py:25-39)
//   re-pad + widen         fourier/profile0.py:51-63   (== profile4.py:48-60)
//   NaN/Inf scrub          fourier/profile0.py:66      (== profile4.py:63)
#pragma once
#include "frad_platform.hpp"

namespace frad {

typedef unsigned long long u64;
// plain compiler vectors for 8/16-byte global accesses (HIP's uint4/double2 are classes and cannot be
// accessed through an address-space qualified pointer)
typedef uint32_t v4u __attribute__((vector_size(16)));
typedef uint32_t v2u __attribute__((vector_size(8)));
typedef double v2d __attribute__((vector_size(16)));

// ---------------------------------------------------------------------------------------------
// bit casts
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(u64 u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ u64 bswap64(u64 v) { return __builtin_bswap64(v); }
__device__ __forceinline__ uint32_t bswap16(uint32_t v) { return ((v & 0xffu) << 8) | ((v >> 8) & 0xffu); }

// ---------------------------------------------------------------------------------------------
// float16 narrowing with a single rounding (numpy's astype('f2') from f8/f4 is correctly rounded)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    return (uint32_t)__half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ uint32_t f64_to_f16_bits(double d) {
    // round-to-odd into f32 (24 significant bits >= 11 + 2), then the hardware RNE f32 -> f16:
    // the pair is a single correctly rounded f64 -> f16 conversion.
    // Round-to-odd = truncate + sticky bit; built from the one-instruction RNE conversion (step back towards zero when
    // it rounded away) instead of a round-towards-zero conversion, which costs ~20 instructions on gfx950.
    const float n = (float)d;
    uint32_t b = f2u(n);
    if ((double)n != d) {                                    // inexact (or NaN: the payload bit is set, as before)
        if (fabs((double)n) > fabs(d)) b -= 1u;               // sign-magnitude: one step towards zero (also from +-inf)
        b |= 1u;
    }
    return f32_to_f16_bits(u2f(b));
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    return __half2float(__ushort_as_half((unsigned short)h));
}

// ---------------------------------------------------------------------------------------------
// R1: one PCM element -> compute type T (double for ints/f64, float for f32/f16)
// code = kind*8 + log2(itemsize)*2 + big_endian   (include/frad_hip.h)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 load_raw(const unsigned char* p, int lg) {
    switch (lg) {
        case 0: return *FRAD_GCPTR(unsigned char, p);
        case 1: return *FRAD_GCPTR(unsigned short, p);
        case 2: return *FRAD_GCPTR(uint32_t, p);
        default: return *FRAD_GCPTR(u64, p);
    }
}

// raw: the element's bytes as loaded little-endian (zero-extended to 64 bit)
template <typename T>
__device__ __forceinline__ T cvt_pcm(u64 raw, int code, bool raw_be_ints) {
    const int kind = code >> 3, lg = (code >> 1) & 3, be = code & 1;
    if (be) {
        if (lg == 1) raw = bswap16((uint32_t)raw);
        else if (lg == 2) raw = bswap32((uint32_t)raw);
        else if (lg == 3) raw = bswap64(raw);
    }
    if (kind == 2) {                                   // floats pass through un-normalised
        if (lg == 1) return (T)f16_bits_to_f32((uint32_t)raw);
        if (lg == 2) return (T)u2f((uint32_t)raw);
        return (T)u2d(raw);
    }
    const int w = 8 << lg;
    double v;
    if (kind == 1) {                                   // signed: sign-extend, RN to f64
        long long s = (long long)(raw << (64 - w)) >> (64 - w);
        v = (double)s;
    } else {
        v = (double)raw;
    }
    if (be && raw_be_ints) return (T)v;                // reference quirk: no scaling at all
    // divide by 2^(w-1) (exact), unsigned then subtract one (one rounding, as numpy does)
    v = v * u2d((u64)(1023 - (w - 1)) << 52);
    if (kind == 0) v = v - 1.0;
    return (T)v;
}

// The same conversion with the format known at compile time (CODE = FRAD_PCM_* value, RAW = the
// big-endian-integer quirk): branch-free, a handful of VALU ops per element.  Kernels pick the
// instantiation once per stage through dispatch_pcm() instead of branching per element.
// NOSCALE: leave signed integers un-normalised -- the caller folds the exact factor 2^-(w-1) into a later
// multiplication (see pcm_deferred_scale), which saves one float64 op per sample.
template <typename T, int CODE, bool RAW, bool NOSCALE = false>
__device__ __forceinline__ T cvt_pcm_c(u64 raw) {
    constexpr int kind = CODE >> 3, lg = (CODE >> 1) & 3, be = CODE & 1;
    if constexpr (be) {
        if constexpr (lg == 1) raw = bswap16((uint32_t)raw);
        else if constexpr (lg == 2) raw = bswap32((uint32_t)raw);
        else if constexpr (lg == 3) raw = bswap64(raw);
    }
    if constexpr (kind == 2) {
        if constexpr (lg == 1) return (T)f16_bits_to_f32((uint32_t)raw);
        else if constexpr (lg == 2) return (T)u2f((uint32_t)raw);
        else return (T)u2d(raw);
    } else {
        constexpr int w = 8 << lg;
        double v;
        if constexpr (kind == 1) {
            if constexpr (lg <= 2) v = (double)((int)((uint32_t)raw << (32 - w)) >> (32 - w));
            else v = (double)(long long)raw;
        } else {
            if constexpr (lg <= 2) v = (double)(uint32_t)raw;
            else v = (double)raw;
        }
        if constexpr (be && RAW) return (T)v;
        if constexpr (NOSCALE && kind == 1) return (T)v;
        v = v * u2d((u64)(1023 - (w - 1)) << 52);
        if constexpr (kind == 0) v = v - 1.0;
        return (T)v;
    }
}
// the factor cvt_pcm_c<.., NOSCALE = true> leaves out for this (dtype, raw_be); 1 when it left nothing out
__device__ __forceinline__ double pcm_deferred_scale(int dtype, int raw_be) {
    const int kind = dtype >> 3, lg = (dtype >> 1) & 3, be = dtype & 1;
    if (kind != 1 || (be && raw_be)) return 1.0;
    return u2d((u64)(1023 - ((8 << lg) - 1)) << 52);
}

template <int V> struct ic { static constexpr int value = V; };
// f(ic<CODE>{}, ic<RAW>{}) for the runtime (dtype, raw_be) pair; LG fixes the item size.
template <int LG, typename F>
__device__ __forceinline__ void dispatch_pcm(int dtype, int raw_be, F&& f) {
    const int kind = dtype >> 3, be = dtype & 1;
    if (kind == 2) {
        if constexpr (LG >= 1) { if (be) f(ic<16 + LG * 2 + 1>{}, ic<0>{}); else f(ic<16 + LG * 2>{}, ic<0>{}); }
    } else if (kind == 1) {
        if constexpr (LG >= 1) {
            if (be) { if (raw_be) f(ic<8 + LG * 2 + 1>{}, ic<1>{}); else f(ic<8 + LG * 2 + 1>{}, ic<0>{}); }
            else f(ic<8 + LG * 2>{}, ic<0>{});
        } else f(ic<8>{}, ic<0>{});
    } else {
        if constexpr (LG >= 1) {
            if (be) { if (raw_be) f(ic<LG * 2 + 1>{}, ic<1>{}); else f(ic<LG * 2 + 1>{}, ic<0>{}); }
            else f(ic<LG * 2>{}, ic<0>{});
        } else f(ic<0>{}, ic<0>{});
    }
}

// ---------------------------------------------------------------------------------------------
// storage codes.  code = the `bits` stored bits of one value, right-aligned, MSB-first order.
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ u64 storage_code(T v, int bits);
template <> __device__ __forceinline__ u64 storage_code<double>(double v, int bits) {
    switch (bits) {
        case 64: return d2u(v);
        case 48: return d2u(v) >> 16;
        case 32: return f2u((float)v);
        case 24: return f2u((float)v) >> 8;
        case 16: return f64_to_f16_bits(v);
        default: return f64_to_f16_bits(v) >> 4;
    }
}
template <> __device__ __forceinline__ u64 storage_code<float>(float v, int bits) {
    switch (bits) {
        case 64: return d2u((double)v);
        case 48: return d2u((double)v) >> 16;
        case 32: return f2u(v);
        case 24: return f2u(v) >> 8;
        case 16: return f32_to_f16_bits(v);
        default: return f32_to_f16_bits(v) >> 4;
    }
}

// inverse: stored code -> float64, then the reference's NaN/Inf -> 0 scrub
__device__ __forceinline__ double code_to_f64(u64 code, int bits) {
    double v;
    switch (bits) {
        case 64: v = u2d(code); break;
        case 48: v = u2d(code << 16); break;
        case 32: v = (double)u2f((uint32_t)code); break;
        case 24: v = (double)u2f((uint32_t)code << 8); break;
        case 16: v = (double)f16_bits_to_f32((uint32_t)code); break;
        default: v = (double)f16_bits_to_f32((uint32_t)code << 4); break;
    }
    // isfinite: exponent field not all ones
    return ((d2u(v) >> 52) & 0x7ff) == 0x7ff ? 0.0 : v;
}

// A "unit" is the smallest run of values whose packed size is a whole number of 16-byte lines:
// bits  12  16  24  32  48  64
// U     32   8  16   4   8   2   values
// bytes 48  16  48  16  48  16
__host__ __device__ constexpr int unit_values(int bits) {
    return bits == 12 ? 32 : bits == 16 ? 8 : bits == 24 ? 16 : bits == 32 ? 4 : bits == 48 ? 8 : 2;
}
__host__ __device__ constexpr int unit_bytes(int bits) { return (bits % 3 == 0) ? 48 : 16; }

// Byte `j` (0-based, stream order) of the packed representation of a value with storage code
// `code`: big-endian = MSB first; little-endian (bits % 8 == 0 only) = LSB first.
__device__ __forceinline__ uint32_t code_byte(u64 code, int bits, bool le, int j) {
    const int nb = bits >> 3;
    const int sh = le ? 8 * j : 8 * (nb - 1 - j);
    return (uint32_t)(code >> sh) & 0xffu;
}

// Pack the storage codes of one unit into `out` words (memory order: out[0]'s low byte is the
// first payload byte).  BITS is a compile-time constant so that every array index is static and
// the unit lives in registers.
template <int BITS>
__device__ __forceinline__ void pack_unit(const u64 (&codes)[unit_values(BITS)], bool le,
                                          uint32_t (&out)[unit_bytes(BITS) / 4]) {
    if constexpr (BITS == 64) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u64 c = le ? codes[i] : bswap64(codes[i]);
            out[2 * i] = (uint32_t)c; out[2 * i + 1] = (uint32_t)(c >> 32);
        }
    } else if constexpr (BITS == 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = le ? (uint32_t)codes[i] : bswap32((uint32_t)codes[i]);
    } else if constexpr (BITS == 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t a = (uint32_t)codes[2 * i], b = (uint32_t)codes[2 * i + 1];
            if (!le) { a = bswap16(a); b = bswap16(b); }
            out[i] = a | (b << 16);
        }
    } else if constexpr (BITS == 12) {
        // always big-endian: two 12-bit codes -> 3 bytes  aaaaaaaa aaaabbbb bbbbbbbb ;
        // eight codes -> 12 bytes -> 3 words
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t by[12];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t a = (uint32_t)codes[g * 8 + 2 * p], b = (uint32_t)codes[g * 8 + 2 * p + 1];
                by[3 * p] = a >> 4; by[3 * p + 1] = ((a & 0xf) << 4) | (b >> 8); by[3 * p + 2] = b & 0xff;
            }
#pragma unroll
            for (int w = 0; w < 3; ++w)
                out[g * 3 + w] = by[4 * w] | (by[4 * w + 1] << 8) | (by[4 * w + 2] << 16) | (by[4 * w + 3] << 24);
        }
    } else {
        // 24 / 48: nb-byte groups back to back
        constexpr int nb = BITS >> 3, U = 48 / nb;
#pragma unroll
        for (int w = 0; w < 12; ++w) out[w] = 0;
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int j = 0; j < nb; ++j) {
                const int s = i * nb + j;
                out[s >> 2] |= code_byte(codes[i], BITS, le, j) << (8 * (s & 3));
            }
    }
}

// Inverse of pack_unit: words (memory order) -> U storage codes.
template <int BITS>
__device__ __forceinline__ void unpack_unit(const uint32_t (&in)[unit_bytes(BITS) / 4], bool le,
                                            u64 (&codes)[unit_values(BITS)]) {
    if constexpr (BITS == 64) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u64 c = (u64)in[2 * i] | ((u64)in[2 * i + 1] << 32);
            codes[i] = le ? c : bswap64(c);
        }
    } else if constexpr (BITS == 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) codes[i] = le ? in[i] : bswap32(in[i]);
    } else if constexpr (BITS == 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t a = in[i] & 0xffffu, b = in[i] >> 16;
            if (!le) { a = bswap16(a); b = bswap16(b); }
            codes[2 * i] = a; codes[2 * i + 1] = b;
        }
    } else if constexpr (BITS == 12) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t by[12];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                by[4 * w] = in[g * 3 + w] & 0xff; by[4 * w + 1] = (in[g * 3 + w] >> 8) & 0xff;
                by[4 * w + 2] = (in[g * 3 + w] >> 16) & 0xff; by[4 * w + 3] = in[g * 3 + w] >> 24;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                codes[g * 8 + 2 * p] = (by[3 * p] << 4) | (by[3 * p + 1] >> 4);
                codes[g * 8 + 2 * p + 1] = ((by[3 * p + 1] & 0xf) << 8) | by[3 * p + 2];
            }
        }
    } else {
        constexpr int nb = BITS >> 3, U = 48 / nb;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            u64 c = 0;
#pragma unroll
            for (int j = 0; j < nb; ++j) {
                const int s = i * nb + j;
                const u64 b = (in[s >> 2] >> (8 * (s & 3))) & 0xffu;
                c |= b << (le ? 8 * j : 8 * (nb - 1 - j));
            }
            codes[i] = c;
        }
    }
}

// Code of value `i` from raw payload bytes, for ragged tails (byte loads, any alignment).
__device__ __forceinline__ u64 code_from_bytes(const unsigned char* frame, int64_t i, int bits, bool le) {
    if (bits == 12) {
        const int64_t nib = 3 * i, b0 = nib >> 1;
        const uint32_t x = frame[b0], y = frame[b0 + 1];
        return (nib & 1) ? (((x & 0xf) << 8) | y) : ((x << 4) | (y >> 4));
    }
    const int nb = bits >> 3;
    u64 c = 0;
    for (int j = 0; j < nb; ++j) c |= (u64)frame[i * nb + j] << (le ? 8 * j : 8 * (nb - 1 - j));
    return c;
}

// byte `s` (stream position inside the frame payload) for the ragged-tail slow path, given a
// functor code_of(i) returning the storage code of value i (or 0 past the end).
template <typename F>
__device__ __forceinline__ uint32_t payload_byte(int64_t s, int bits, bool le, int64_t n_values, F code_of) {
    if (bits == 12) {
        const int64_t nib = 2 * s;                       // first nibble of this byte
        uint32_t out = 0;
        for (int h = 0; h < 2; ++h) {
            const int64_t q = nib + h, i = q / 3; const int r = (int)(q - 3 * i);
            const uint32_t c = i < n_values ? (uint32_t)code_of(i) : 0u;
            out = (out << 4) | ((c >> (4 * (2 - r))) & 0xfu);
        }
        return out;
    }
    const int nb = bits >> 3;
    const int64_t i = s / nb; const int j = (int)(s - i * nb);
    return code_byte(code_of(i), bits, le, j);
}

// ---------------------------------------------------------------------------------------------
// |x| max with numpy semantics: the bit pattern of |x| is monotone in the value for non-NaN and
// every NaN pattern is larger than +Inf, so an unsigned max reproduces np.max(np.abs(.)) incl.
// "any NaN -> NaN".
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 abs_bits(double v) { return d2u(v) & 0x7fffffffffffffffULL; }

// Wave-wide butterfly all-reduce (xor 1, 2, 4, 8, 16, 32) of a 64-bit value.  On gfx950 the first four steps are DPP
// moves -- quad_perm for xor 1 / 2, row_half_mirror and row_mirror for xor 4 / 8 (their partners hold the same value as
// the xor partners once the smaller groups are uniform) -- and the last two read one lane per 16-lane row through
// SGPRs, so no step goes through the LDS crossbar (a ds_bpermute pair per step otherwise).  The emulator runs the
// plain xor butterfly; both combine the same operands in the same order.
#ifndef FRAD_HOST_EMULATION
template <int CTRL> __device__ __forceinline__ u64 dpp_move_u64(u64 v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return (u64)(uint32_t)lo | ((u64)(uint32_t)hi << 32);
}
__device__ __forceinline__ u64 read_lane_u64(u64 v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (u64)lo | ((u64)hi << 32);
}
#endif
template <class Op> __device__ __forceinline__ u64 wave_allreduce_u64(u64 v, Op op) {
#ifdef FRAD_HOST_EMULATION
    for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    return v;
#else
    v = op(v, dpp_move_u64<0xB1>(v));                        // quad_perm [1,0,3,2]
    v = op(v, dpp_move_u64<0x4E>(v));                        // quad_perm [2,3,0,1]
    v = op(v, dpp_move_u64<0x141>(v));                       // row_half_mirror
    v = op(v, dpp_move_u64<0x140>(v));                       // row_mirror
    const u64 r0 = read_lane_u64(v, 0), r1 = read_lane_u64(v, 16), r2 = read_lane_u64(v, 32), r3 = read_lane_u64(v, 48);
    return op(op(r0, r1), op(r2, r3));
#endif
}

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
    return wave_allreduce_u64(v, [](u64 a, u64 b) { return b > a ? b : a; });
}

// ---------------------------------------------------------------------------------------------
// decoder output conversion: one float64 sample -> the bytes of PCM format (KIND, log2 itemsize), little-endian in the
// returned word.  from_f64(pcm, fmt).astype(fmt) of the reference's caller (backend/pcmformat.py:49-62, src/decoder.py:23);
// integer overflow as numpy yields it on x86-64 (cvttsd2si), see frad_epilogue.hip.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t x86_cvtt32(double v) {      // cvttsd2si r32
    return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
}
__device__ __forceinline__ long long x86_cvtt64(double v) {    // cvttsd2si r64
    return (v >= -9223372036854775808.0 && v < 9223372036854775808.0) ? (long long)v : (long long)0x8000000000000000ull;
}

// one float64 sample -> the element's bytes (little-endian in the returned word; `be` formats are swapped by the caller)
// `raw`: the reference's big-endian-integer quirk on the way out -- pcm_format == np.int16 is False for '>i2', so
// from_f64 returns the float64 samples unscaled and the caller's .astype(fmt) truncates those (FRAD_RAW_BE_INTS)
template <int KIND, int LGS>
__device__ __forceinline__ u64 from_f64_bits(double x, bool raw) {
    if constexpr (KIND == 2) {
        if constexpr (LGS == 1) return f64_to_f16_bits(x);
        else if constexpr (LGS == 2) return f2u((float)x);
        else return d2u(x);
    } else {
        constexpr int w = 8 << LGS;
        const double scale = u2d((u64)(1023 + (w - 1)) << 52);          // 2^(w-1), exact
        const double v = raw ? x : (KIND == 0 ? x + 1.0 : x) * scale;
        if constexpr (LGS <= 1) return (u64)((uint32_t)x86_cvtt32(v) & ((1u << w) - 1u));
        else if constexpr (LGS == 2) return KIND == 1 ? (u64)(uint32_t)x86_cvtt32(v) : (u64)(uint32_t)x86_cvtt64(v);
        else {
            if constexpr (KIND == 1) return (u64)x86_cvtt64(v);
            else return v >= 9223372036854775808.0 ? (u64)x86_cvtt64(v - 9223372036854775808.0) + 0x8000000000000000ull : (u64)x86_cvtt64(v);
        }
    }
}
}  // namespace frad
