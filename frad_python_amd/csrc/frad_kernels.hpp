// frad_kernels.hpp -- the HIP kernels of the FrAD transform core (gfx950 / MI355X, wave64).
//
//   K1  k_p4_pack      to_f64 + absmax + cast + bit-depth pack            (profile4.analogue)
//   K2  k_p4_unpack    unpack + widen + NaN/Inf scrub                      (profile4.digital)
//   K3/K5 k_p0_fwd     fused load -> to_f64 -> DCT-II (f64 or f32) -> absmax -> cast -> pack
//                                                                         (profile0.analogue)
//   K4  k_p0_inv       fused unpack -> scrub -> inverse DCT (f64) -> interleave -> store
//                                                                         (profile0.digital)
//   direct variants    any frame length N (tails, odd sizes): O(N^2) cosine sums from an exact
//                      table, same load / pack stages.
// Reference line numbers are in include/frad_hip.h and DESIGN.md.
#pragma once
#include "frad_common.hpp"
#include "frad_fft.hpp"

namespace frad {

struct Geom {
    long long n_frames;
    long long frame_stride;     // sample-frames between consecutive frames of the PCM input
    long long payload_stride;   // bytes between consecutive frame payloads
    int N, C;                   // sample-frames per frame, channels
    int bits;                   // storage depth 12/16/24/32/48/64
    int le;                     // payload little-endian (ignored for 12 bit)
    int dtype;                  // FRAD_PCM_* code of the PCM side (encode only)
    int raw_be;                 // reference quirk: big-endian ints are not normalised
    int fpb;                    // frames per block (p0 kernels)
    int n_valid;                // sample-frames actually read per frame (<= N, rest zero; profile 1)
};

__device__ __forceinline__ bool dtype_is_f32_class(int code) { return (code >> 3) == 2 && ((code >> 1) & 3) <= 2; }

template <int NW> __device__ __forceinline__ void load_words(const unsigned char* p, uint32_t (&w)[NW]) {
    if constexpr (NW >= 4) {
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else if constexpr (NW == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y;
    } else {
        w[0] = *reinterpret_cast<const uint32_t*>(p);
    }
}
template <int NW> __device__ __forceinline__ void store_words(unsigned char* p, const uint32_t (&w)[NW]) {
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
        uint4 v; v.x = w[4 * i]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3];
        reinterpret_cast<uint4*>(p)[i] = v;
    }
}
// element i (LG = log2 itemsize) of a little-endian word array
template <int LG, int NW> __device__ __forceinline__ u64 word_elem(const uint32_t (&w)[NW], int i) {
    if constexpr (LG == 0) return (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
    else if constexpr (LG == 1) return (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
    else if constexpr (LG == 2) return w[i];
    else return (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
}

__device__ __forceinline__ void block_absmax_commit(u64 mx, double* absmax, long long f) {
    mx = wave_max_u64(mx);
    if (absmax != nullptr && (threadIdx.x & 63) == 0 && mx != 0)
        atomicMax(reinterpret_cast<u64*>(absmax) + f, mx);
}

// =============================================================================================
// K1  profile 4 encode, aligned fast path: one thread = one pack unit.
// grid.x = n_frames * bpf ; requires NC >= U, pcm/payload bases and strides 16-byte aligned.
// The < U ragged values at the end of a frame (NC % U) are finished byte-wise by the frame's
// first block.
// =============================================================================================
template <typename T, int BITS, int LG>
__device__ __forceinline__ void p4_pack_body(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                             double* absmax, const Geom& g, int bpf) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int RAWB = U << LG, NW = RAWB >= 4 ? RAWB / 4 : 1;
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C;
    const long long units = NC / U;
    const unsigned char* src = pcm + ((f * g.frame_stride * g.C) << LG);
    unsigned char* dst = payload + f * g.payload_stride;
    const bool le = g.le && (BITS % 8 == 0);
    u64 mx = 0;
    for (long long u = (long long)chunk * blockDim.x + threadIdx.x; u < units; u += (long long)bpf * blockDim.x) {
        uint32_t w[NW];
        if constexpr (RAWB >= 4) load_words<NW>(src + u * RAWB, w);
        else w[0] = *reinterpret_cast<const unsigned short*>(src + u * RAWB);
        u64 codes[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const T v = cvt_pcm<T>(word_elem<LG>(w, i), g.dtype, g.raw_be);
            const u64 a = abs_bits((double)v);
            mx = a > mx ? a : mx;
            codes[i] = storage_code<T>(v, BITS);
        }
        uint32_t out[UB / 4];
        pack_unit<BITS>(codes, le, out);
        store_words<UB / 4>(dst + u * UB, out);
    }
    const int tail = (int)(NC - units * U);
    if (tail && chunk == 0) {
        const long long first = units * U;
        auto value = [&](long long i) { return cvt_pcm<T>(load_raw(src + (i << LG), LG), g.dtype, g.raw_be); };
        auto code_of = [&](long long i) -> u64 { return i < NC ? storage_code<T>(value(i), BITS) : 0; };
        const long long b0 = units * UB, b1 = (BITS == 12) ? (NC * 3 + 1) / 2 : NC * (BITS / 8);
        for (long long s = b0 + threadIdx.x; s < b1; s += blockDim.x)
            dst[s] = (unsigned char)payload_byte(s, BITS, le, NC, code_of);
        for (long long i = first + threadIdx.x; i < NC; i += blockDim.x) {
            const u64 a = abs_bits((double)value(i));
            mx = a > mx ? a : mx;
        }
    }
    block_absmax_commit(mx, absmax, f);
}

template <int BITS, int LG>
__global__ void __launch_bounds__(256) k_p4_pack(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                 double* absmax, Geom g, int bpf) {
    if (dtype_is_f32_class(g.dtype)) p4_pack_body<float, BITS, LG>(pcm, payload, absmax, g, bpf);
    else p4_pack_body<double, BITS, LG>(pcm, payload, absmax, g, bpf);
}

// Any alignment / any geometry: one thread per payload byte and per value.
template <int UNUSED>
__global__ void __launch_bounds__(256) k_p4_pack_slow(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                      double* absmax, Geom g, int bpf) {
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C;
    const int lg = (g.dtype >> 1) & 3, bits = g.bits;
    const bool le = g.le && (bits % 8 == 0), f32c = dtype_is_f32_class(g.dtype);
    const unsigned char* src = pcm + ((f * g.frame_stride * g.C) << lg);
    unsigned char* dst = payload + f * g.payload_stride;
    auto value = [&](long long i) -> double {
        if (f32c) return (double)cvt_pcm<float>(load_raw(src + (i << lg), lg), g.dtype, g.raw_be);
        return cvt_pcm<double>(load_raw(src + (i << lg), lg), g.dtype, g.raw_be);
    };
    auto code_of = [&](long long i) -> u64 {
        if (i >= NC) return 0;
        return f32c ? storage_code<float>((float)value(i), bits) : storage_code<double>(value(i), bits);
    };
    const long long nbytes = (bits == 12) ? (NC * 3 + 1) / 2 : NC * (bits / 8);
    u64 mx = 0;
    for (long long s = (long long)chunk * blockDim.x + threadIdx.x; s < nbytes; s += (long long)bpf * blockDim.x)
        dst[s] = (unsigned char)payload_byte(s, bits, le, NC, code_of);
    for (long long i = (long long)chunk * blockDim.x + threadIdx.x; i < NC; i += (long long)bpf * blockDim.x) {
        const u64 a = abs_bits(value(i));
        mx = a > mx ? a : mx;
    }
    block_absmax_commit(mx, absmax, f);
}

// =============================================================================================
// K2  profile 4 decode.  Fast: one thread = one unit (aligned); slow: one thread = one value.
// =============================================================================================
template <int BITS>
__global__ void __launch_bounds__(256) k_p4_unpack(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                   Geom g, int bpf) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C;
    const long long units = NC / U;
    const unsigned char* src = payload + f * g.payload_stride;
    double* dst = out + f * NC;
    const bool le = g.le && (BITS % 8 == 0);
    for (long long u = (long long)chunk * blockDim.x + threadIdx.x; u < units; u += (long long)bpf * blockDim.x) {
        uint32_t w[UB / 4];
        load_words<UB / 4>(src + u * UB, w);
        u64 codes[U];
        unpack_unit<BITS>(w, le, codes);
#pragma unroll
        for (int i = 0; i < U / 2; ++i) {
            double2 v; v.x = code_to_f64(codes[2 * i], BITS); v.y = code_to_f64(codes[2 * i + 1], BITS);
            reinterpret_cast<double2*>(dst + u * U)[i] = v;
        }
    }
    if (chunk == 0)
        for (long long i = units * U + threadIdx.x; i < NC; i += blockDim.x)
            dst[i] = code_to_f64(code_from_bytes(src, i, BITS, le), BITS);
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p4_unpack_slow(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                        Geom g, int bpf) {
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C;
    const unsigned char* src = payload + f * g.payload_stride;
    const bool le = g.le && (g.bits % 8 == 0);
    for (long long i = (long long)chunk * blockDim.x + threadIdx.x; i < NC; i += (long long)bpf * blockDim.x)
        out[f * NC + i] = code_to_f64(code_from_bytes(src, i, g.bits, le), g.bits);
}

// =============================================================================================
// shared stages of the profile-0 kernels.  A block owns `fpb` consecutive frames with all their
// channels; channel-frame cf = fl * C + c (fl = frame inside the block) has its own LDS buffer of
// SLOTS complex slots.  `XS` (slot stride) differs between the FFT kernels (padded complex
// layout, reals overlaid) and the direct kernels (plain real arrays).
// =============================================================================================

// value of real slot r of channel-frame cf
template <typename T, bool PADDED>
__device__ __forceinline__ T& xslot(unsigned char* smem, int cf, int slots, int r) {
    if constexpr (PADDED) return real_slot(reinterpret_cast<cx<T>*>(smem) + (long long)cf * slots, r);
    else return reinterpret_cast<T*>(smem)[(long long)cf * slots + r];
}

// Stage-in (encode): interleaved PCM of the block's frames -> T in LDS.  PERMUTE applies Makhoul's
// even/odd permutation (FFT kernels); the direct kernels keep time order.
template <typename T, int LG, bool PADDED, bool PERMUTE>
__device__ __forceinline__ void stage_in_pcm(const unsigned char* __restrict__ pcm, unsigned char* smem, const Geom& g,
                                             long long f0, int nfl, int slots, bool aligned) {
    const int N = g.N, C = g.C, NC = N * C;
    const int nv = g.n_valid * C;                     // elements actually present per frame
    if (aligned) {
        constexpr int EPC = 16 >> LG;                 // elements per 16-byte chunk
        const int chunks = (NC + EPC - 1) / EPC;      // NC * itemsize is a multiple of 16 here
        for (int q = threadIdx.x; q < nfl * chunks; q += blockDim.x) {
            const int fl = q / chunks, ch = q - fl * chunks;
            const unsigned char* src = pcm + (((f0 + fl) * g.frame_stride * C) << LG);
            uint32_t w[4];
            const int e0 = ch * EPC;
            if (e0 + EPC <= nv) load_words<4>(src + ((long long)e0 << LG), w);
            else {
                w[0] = w[1] = w[2] = w[3] = 0;
                for (int i = 0; i < EPC && e0 + i < nv; ++i) {   // ragged end of a short frame
                    const u64 r = load_raw(src + ((long long)(e0 + i) << LG), LG);
                    if constexpr (LG == 3) { w[2 * i] = (uint32_t)r; w[2 * i + 1] = (uint32_t)(r >> 32); }
                    else if constexpr (LG == 2) w[i] = (uint32_t)r;
                    else if constexpr (LG == 1) w[i >> 1] |= (uint32_t)r << (16 * (i & 1));
                    else w[i >> 2] |= (uint32_t)r << (8 * (i & 3));
                }
            }
            int n = e0 / C, c = e0 - n * C;
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int e = e0 + i;
                if (e < NC) {
                    const T v = e < nv ? cvt_pcm<T>(word_elem<LG>(w, i), g.dtype, g.raw_be) : (T)0;
                    xslot<T, PADDED>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n) = v;
                }
                if (++c == C) { c = 0; ++n; }
            }
        }
    } else {
        for (int q = threadIdx.x; q < nfl * NC; q += blockDim.x) {
            const int fl = q / NC, e = q - fl * NC;
            const int n = e / C, c = e - n * C;
            const unsigned char* src = pcm + (((f0 + fl) * g.frame_stride * C) << LG);
            const T v = e < nv ? cvt_pcm<T>(load_raw(src + ((long long)e << LG), LG), g.dtype, g.raw_be) : (T)0;
            xslot<T, PADDED>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n) = v;
        }
    }
}

// Epilogue (encode): X[k] of every channel-frame in LDS -> absmax, storage cast, pack, store.
// Payload order is bin-major / channel-minor (profile0.py:30: freqs.T.ravel()).
template <typename T, int BITS, bool PADDED>
__device__ __forceinline__ void pack_out(unsigned char* smem, unsigned char* __restrict__ payload, double* absmax,
                                         const Geom& g, long long f0, int nfl, int slots, bool aligned) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    const int C = g.C, NC = g.N * C;
    const bool le = g.le && (BITS % 8 == 0);
    const int units = aligned ? NC / U : 0;
    const long long nbytes = (BITS == 12) ? ((long long)NC * 3 + 1) / 2 : (long long)NC * (BITS / 8);
    for (int fl = 0; fl < nfl; ++fl) {
        unsigned char* dst = payload + (f0 + fl) * g.payload_stride;
        u64 mx = 0;
        for (int u = threadIdx.x; u < units; u += blockDim.x) {
            u64 codes[U];
            int k = (u * U) / C, c = (u * U) - k * C;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const T v = xslot<T, PADDED>(smem, fl * C + c, slots, k);
                const u64 a = abs_bits((double)v);
                mx = a > mx ? a : mx;
                codes[i] = storage_code<T>(v, BITS);
                if (++c == C) { c = 0; ++k; }
            }
            uint32_t out[UB / 4];
            pack_unit<BITS>(codes, le, out);
            store_words<UB / 4>(dst + (long long)u * UB, out);
        }
        if (units * U < NC) {                          // ragged tail or unaligned payload: byte-wise
            auto value = [&](long long i) -> T { const int k = (int)(i / C); return xslot<T, PADDED>(smem, fl * C + (int)(i - (long long)k * C), slots, k); };
            auto code_of = [&](long long i) -> u64 { return i < NC ? storage_code<T>(value(i), BITS) : 0; };
            for (long long s = (long long)units * UB + threadIdx.x; s < nbytes; s += blockDim.x)
                dst[s] = (unsigned char)payload_byte(s, BITS, le, NC, code_of);
            for (int i = units * U + threadIdx.x; i < NC; i += blockDim.x) {
                const u64 a = abs_bits((double)value(i));
                mx = a > mx ? a : mx;
            }
        }
        block_absmax_commit(mx, absmax, f0 + fl);
    }
}

template <typename T, bool PADDED>
__device__ __forceinline__ void pack_out_any(unsigned char* smem, unsigned char* __restrict__ payload, double* absmax,
                                             const Geom& g, long long f0, int nfl, int slots, bool aligned) {
    switch (g.bits) {
        case 12: pack_out<T, 12, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 16: pack_out<T, 16, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 24: pack_out<T, 24, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 32: pack_out<T, 32, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 48: pack_out<T, 48, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        default: pack_out<T, 64, PADDED>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
    }
}

// Stage-in (decode): payload -> unpack -> scrub -> X[k] (float64) in LDS.
template <int BITS, bool PADDED>
__device__ __forceinline__ void unpack_in(const unsigned char* __restrict__ payload, unsigned char* smem, const Geom& g,
                                          long long f0, int nfl, int slots, bool aligned) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    const int C = g.C, NC = g.N * C;
    const bool le = g.le && (BITS % 8 == 0);
    const int units = aligned ? NC / U : 0;
    for (int fl = 0; fl < nfl; ++fl) {
        const unsigned char* src = payload + (f0 + fl) * g.payload_stride;
        for (int u = threadIdx.x; u < units; u += blockDim.x) {
            uint32_t w[UB / 4];
            load_words<UB / 4>(src + (long long)u * UB, w);
            u64 codes[U];
            unpack_unit<BITS>(w, le, codes);
            int k = (u * U) / C, c = (u * U) - k * C;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                xslot<double, PADDED>(smem, fl * C + c, slots, k) = code_to_f64(codes[i], BITS);
                if (++c == C) { c = 0; ++k; }
            }
        }
        for (int i = units * U + threadIdx.x; i < NC; i += blockDim.x) {
            const int k = i / C, c = i - k * C;
            xslot<double, PADDED>(smem, fl * C + c, slots, k) = code_to_f64(code_from_bytes(src, i, BITS, le), BITS);
        }
    }
}
template <bool PADDED>
__device__ __forceinline__ void unpack_in_any(const unsigned char* __restrict__ payload, unsigned char* smem, const Geom& g,
                                              long long f0, int nfl, int slots, bool aligned) {
    switch (g.bits) {
        case 12: unpack_in<12, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 16: unpack_in<16, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 24: unpack_in<24, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 32: unpack_in<32, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 48: unpack_in<48, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
        default: unpack_in<64, PADDED>(payload, smem, g, f0, nfl, slots, aligned); break;
    }
}

// Epilogue (decode): time samples in LDS -> interleaved float64 [N, C] rows, 16 bytes per lane.
template <bool PADDED, bool PERMUTE>
__device__ __forceinline__ void store_pcm_f64(unsigned char* smem, double* __restrict__ out, const Geom& g,
                                              long long f0, int nfl, int slots) {
    const int N = g.N, C = g.C, NC = N * C;
    const int pairs = NC / 2;                       // out + f*NC is 16-byte aligned when NC is even
    for (int fl = 0; fl < nfl; ++fl) {
        double* dst = out + (f0 + fl) * (long long)NC;
        if ((NC & 1) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            for (int p = threadIdx.x; p < pairs; p += blockDim.x) {
                const int e = 2 * p;
                int n = e / C, c = e - n * C;
                double2 v;
                v.x = xslot<double, PADDED>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
                if (++c == C) { c = 0; ++n; }
                v.y = xslot<double, PADDED>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
                reinterpret_cast<double2*>(dst)[p] = v;
            }
        } else {
            for (int e = threadIdx.x; e < NC; e += blockDim.x) {
                const int n = e / C, c = e - n * C;
                dst[e] = xslot<double, PADDED>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
            }
        }
    }
}

// =============================================================================================
// K3/K5  profile 0 encode, N = 2^(LOG2M+1): LDS-resident FFT.
// block = fpb frames x C channels x TEAM lanes (rounded up to whole waves).
// =============================================================================================
template <typename T, int LOG2M, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_p0_fwd(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                  double* absmax, const cx<T>* __restrict__ tw, const cx<T>* __restrict__ post,
                                                  Geom g, int aligned_in, int aligned_out) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M);
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    stage_in_pcm<T, LG, true, true>(pcm, smem, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    // blockDim.x == fpb * C * TEAM exactly (a whole number of waves): every team owns a buffer.
    // In a short last block the teams of the missing frames transform whatever the LDS holds and
    // nobody reads their result; that keeps every barrier of the multi-wave teams uniform.
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)cf * SLOTS;
    fft_team<T, LOG2M, false>(buf, t, tw);
    dct_post<T, LOG2M>(buf, t, post);
    __syncthreads();
    pack_out_any<T, true>(smem, payload, absmax, g, f0, nfl, SLOTS, aligned_out != 0);
}

// =============================================================================================
// K4  profile 0 decode (always float64, as the reference widens before idct).
// =============================================================================================
template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT) k_p0_inv(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                  const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post,
                                                  Geom g, int aligned_in) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M);
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    unpack_in_any<true>(payload, smem, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    dct_pre_inverse<double, LOG2M>(buf, t, post);
    fft_team<double, LOG2M, true>(buf, t, tw);
    __syncthreads();
    store_pcm_f64<true, true>(smem, out, g, f0, nfl, SLOTS);
}

// =============================================================================================
// direct kernels: any N.  cos table ct[j] = cos(pi * j / (2N)), j in [0, 4N).
// LDS: x (T) and X (T) per channel-frame, plain arrays of N reals each.
// =============================================================================================
template <typename T, int LG>
__global__ void __launch_bounds__(256) k_p0_fwd_direct(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                       double* absmax, const double* __restrict__ ct, Geom g,
                                                       int aligned_in, int aligned_out) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    T* x = reinterpret_cast<T*>(smem);
    T* X = x + (long long)N * C;
    stage_in_pcm<T, LG, false, false>(pcm, smem, g, f0, 1, N, aligned_in != 0);
    __syncthreads();
    const double inv_n = 1.0 / (double)N;
    const unsigned fourN = 4u * (unsigned)N;
    for (int q = threadIdx.x; q < N * C; q += blockDim.x) {
        const int c = q / N, k = q - c * N;
        const T* xc = x + (long long)c * N;
        double acc = 0.0;
        unsigned j = (unsigned)k % fourN;           // k * (2n + 1) mod 4N, advanced by 2k per sample
        const unsigned step = (2u * (unsigned)k) % fourN;
        for (int n = 0; n < N; ++n) {
            acc = fma((double)xc[n], ct[j], acc);
            j += step; if (j >= fourN) j -= fourN;
        }
        X[(long long)c * N + k] = (T)(acc * inv_n);
    }
    __syncthreads();
    pack_out_any<T, false>(reinterpret_cast<unsigned char*>(X), payload, absmax, g, f0, 1, N, aligned_out != 0);
}

template <int UNUSED>
__global__ void __launch_bounds__(256) k_p0_inv_direct(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                       const double* __restrict__ ct, Geom g, int aligned_in) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = blockIdx.x;
    double* X = reinterpret_cast<double*>(smem);
    double* x = X + (long long)N * C;
    unpack_in_any<false>(payload, smem, g, f0, 1, N, aligned_in != 0);
    __syncthreads();
    const unsigned fourN = 4u * (unsigned)N;
    for (int q = threadIdx.x; q < N * C; q += blockDim.x) {
        const int c = q / N, n = q - c * N;
        const double* Xc = X + (long long)c * N;
        double acc = 0.0;
        const unsigned step = (2u * (unsigned)n + 1u) % fourN;
        unsigned j = step;                          // k * (2n + 1) mod 4N for k = 1
        for (int k = 1; k < N; ++k) {
            acc = fma(Xc[k], ct[j], acc);
            j += step; if (j >= fourN) j -= fourN;
        }
        x[(long long)c * N + n] = Xc[0] + 2.0 * acc;
    }
    __syncthreads();
    store_pcm_f64<false, false>(reinterpret_cast<unsigned char*>(x), out, g, f0, 1, N);
}

}  // namespace frad
