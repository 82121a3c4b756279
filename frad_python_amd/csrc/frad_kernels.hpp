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
#include <type_traits>
#include "frad_common.hpp"
#include "frad_fft.hpp"

#ifndef FRAD_NOINLINE
// Stage functions are real calls: each gets its own register allocation, so the rarely used pack
// formats cannot inflate the FFT core's VGPR budget (occupancy: two 256-thread blocks per CU).
#define FRAD_NOINLINE __attribute__((noinline))
#endif

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
    int cg;                     // channels transformed per pass; < C only when one frame's channels
                                // exceed a CU's LDS (then fpb == 1 and I/O is per value)
    int in_mode;                // FFT kernels, PCM side: 0 = per element, 1/2/3 = quad stage-in (row bytes
                                // divide 16 / equal 8 / multiple of 16), see stage_in_quads
    int cc_fast;                // FFT kernels, payload side: 1 or 2 = pairwise 16-byte LDS path for C = 1 / 2
    int* ovf_flag;              // frad_p0_analogue_checked: set to 1 when a frame's |X| max exceeds ovf_limit (else untouched); may be null
    double ovf_limit;
    int fpc;                    // frames per clip of the PCM side ( > 0: frad_p0_*_clips; 0: one flat run of frames)
    long long clip_stride;      // sample-frames between consecutive clips
};
// first sample-frame of frame f on the PCM side: i * frame_stride, or -- a batch of equally cut clips, encoder.py:72-93 per
// clip -- clip * clip_stride + (frame inside the clip) * frame_stride.  (n_frames < 2^31 whenever fpc is set.)
__device__ __forceinline__ long long frame_base(const Geom& g, long long f) {
    if (g.fpc <= 0) return f * g.frame_stride;
    const unsigned c = (unsigned)f / (unsigned)g.fpc;
    return (long long)c * g.clip_stride + (long long)((unsigned)f - c * (unsigned)g.fpc) * g.frame_stride;
}

__device__ __forceinline__ bool dtype_is_f32_class(int code) { return (code >> 3) == 2 && ((code >> 1) & 3) <= 2; }

template <int NW> __device__ __forceinline__ void load_words(const unsigned char* p, uint32_t (&w)[NW]) {
    if constexpr (NW >= 4) {
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const v4u v = FRAD_GCPTR(v4u, p)[i];
            w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3];
        }
    } else if constexpr (NW == 3) {                           // 12 bytes, 4-byte aligned (24- / 48-bit sub-units)
        const auto* q = FRAD_GCPTR(uint32_t, p);
        w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
    } else if constexpr (NW == 2) {
        const v2u v = *FRAD_GCPTR(v2u, p); w[0] = v[0]; w[1] = v[1];
    } else {
        w[0] = *FRAD_GCPTR(uint32_t, p);
    }
}
// STREAM: nontemporal store.  Measured (profiles/r02_bench_extra.jsonl vs r01): it pays only where a wave's store
// instruction covers whole contiguous lines (consecutive lanes, consecutive 16-byte pieces); for lane-strided pieces it
// defeats the write combining in L2 and costs 20-35 %.
template <int NW, bool STREAM = false> __device__ __forceinline__ void store_words(unsigned char* p, const uint32_t (&w)[NW]) {
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
        v4u v = {w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
        if constexpr (STREAM) FRAD_NT_STORE(v, FRAD_GPTR(v4u, p) + i); else *(FRAD_GPTR(v4u, p) + i) = v;
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
                                             double* absmax, const Geom& g, int bpf, v4u* stage) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int RAWB = U << LG, NW = RAWB >= 4 ? RAWB / 4 : 1;
    // bpf > 0: a frame is shared by bpf blocks.  bpf == 0 (frames of fewer than 4 units per thread of a block: the stereo
    // N = 2048 frames of the BASELINE configurations are 512 units at 16 bit): ONE WAVE per frame, so that a lane still has
    // four loads in flight; a block of 256 threads then takes four frames.
    long long f; int chunk, tid, tsz;
    if (bpf > 0) { f = blockIdx.x / bpf; chunk = blockIdx.x - (int)(f * bpf); tid = threadIdx.x; tsz = blockDim.x; }
    else {
        f = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); chunk = 0; tid = threadIdx.x & 63; tsz = 64;
        if (f >= g.n_frames) return;                         // (whole waves leave; the kernel has no block barrier)
    }
    const long long NC = (long long)g.N * g.C;
    const long long units = NC / U;
    const unsigned char* src = pcm + ((f * g.frame_stride * g.C) << LG);
    unsigned char* dst = payload + f * g.payload_stride;
    const bool le = g.le && (BITS % 8 == 0);
    u64 mx = 0;
    // the PCM format is resolved once (dispatch_pcm), not per element; four units are loaded before the first is
    // converted so that 4 x RAWB bytes per lane are in flight
    dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
        constexpr int CODE = decltype(code_tag)::value;
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        auto load = [&](long long u, uint32_t (&w)[NW]) {
            if constexpr (RAWB >= 4) load_words<NW>(src + u * RAWB, w);
            else w[0] = *FRAD_GCPTR(unsigned short, src + u * RAWB);
        };
        const int lane = threadIdx.x & 63;
        auto emit = [&](long long u, const uint32_t (&w)[NW], bool valid) {      // called by whole waves (`valid`: this lane has a unit)
            uint32_t out[UB / 4];
            if (valid) {
                u64 codes[U];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const T v = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(w, i));
                    const u64 a = abs_bits((double)v);
                    mx = a > mx ? a : mx;
                    codes[i] = storage_code<T>(v, BITS);
                }
                pack_unit<BITS>(codes, le, out);
            }
#ifndef FRAD_HOST_EMULATION
            if constexpr (UB == 48) {
                // 48-byte units: a lane's three 16-byte pieces lie 48 bytes apart in the payload, 64 places per store
                // instruction.  A wave that holds 64 consecutive units passes the pieces through LDS (piece 3 lane + k in,
                // piece 64 k + lane out; 12- and 4-bank strides: conflict-free) and stores three whole 1 KiB rows.
                const long long u0 = u - lane;                // the wave's first unit (the same in every lane)
                if (stage != nullptr && u0 + 64 <= units) {
                    v4u* st = stage + (threadIdx.x >> 6) * 192;
#pragma unroll
                    for (int k = 0; k < 3; ++k) st[3 * lane + k] = v4u{out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]};
                    team_sync<64>();
                    v4u row[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) row[k] = st[64 * k + lane];
                    team_sync<64>();
#pragma unroll
                    for (int k = 0; k < 3; ++k) FRAD_NT_STORE(row[k], FRAD_GPTR(v4u, dst + u0 * UB) + 64 * k + lane);
                    return;
                }
            }
#endif
            if (valid) store_words<UB / 4, BITS == 32>(dst + u * UB, out);
        };
        const long long step = bpf > 0 ? (long long)bpf * tsz : tsz;
        long long u = (long long)chunk * tsz + tid;
        // loop conditions on the wave's LAST lane (u - lane + 63), so that a wave stays together.  NBAT units of a thread are
        // loaded before the first is converted: at least 64 raw bytes in flight per lane, but no more units than that takes
        // (four 64-byte units at 12 bit cost the kernel its occupancy: 332 VGPRs)
        constexpr int NBAT = RAWB >= 64 ? 1 : RAWB >= 32 ? 2 : 4;
        for (; u - lane + 63 + (NBAT - 1) * step < units; u += NBAT * step) {
            uint32_t w[NBAT][NW];
#pragma unroll
            for (int b = 0; b < NBAT; ++b) load(u + b * step, w[b]);
#pragma unroll
            for (int b = 0; b < NBAT; ++b) emit(u + b * step, w[b], true);
        }
        for (; u - lane < units; u += NBAT * step) {         // what is left (ragged waves), loaded together as well
            uint32_t w[NBAT][NW];
#pragma unroll
            for (int b = 0; b < NBAT; ++b) if (u + b * step < units) load(u + b * step, w[b]);
#pragma unroll
            for (int b = 0; b < NBAT; ++b) if (u - lane + b * step < units) emit(u + b * step, w[b], u + b * step < units);
        }
    });
    const int tail = (int)(NC - units * U);
    if (tail && chunk == 0) {
        const long long first = units * U;
        auto value = [&](long long i) { return cvt_pcm<T>(load_raw(src + (i << LG), LG), g.dtype, g.raw_be); };
        auto code_of = [&](long long i) -> u64 { return i < NC ? storage_code<T>(value(i), BITS) : 0; };
        const long long b0 = units * UB, b1 = (BITS == 12) ? (NC * 3 + 1) / 2 : NC * (BITS / 8);
        for (long long s = b0 + tid; s < b1; s += tsz)
            dst[s] = (unsigned char)payload_byte(s, BITS, le, NC, code_of);
        for (long long i = first + tid; i < NC; i += tsz) {
            const u64 a = abs_bits((double)value(i));
            mx = a > mx ? a : mx;
        }
    }
    block_absmax_commit(mx, absmax, f);
}

template <int BITS, int LG>
__global__ void __launch_bounds__(256) k_p4_pack(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                 double* absmax, Geom g, int bpf) {
    v4u* stage = nullptr;
#ifndef FRAD_HOST_EMULATION
    if constexpr (unit_bytes(BITS) == 48) {                  // 3 KiB per wave: the 48-byte units' pieces change lanes here
        __shared__ v4u p4_stage[4 * 192];
        stage = p4_stage;
    }
#endif
    if (dtype_is_f32_class(g.dtype)) p4_pack_body<float, BITS, LG>(pcm, payload, absmax, g, bpf, stage);
    else p4_pack_body<double, BITS, LG>(pcm, payload, absmax, g, bpf, stage);
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
            v2d v = {code_to_f64(codes[2 * i], BITS), code_to_f64(codes[2 * i + 1], BITS)};
            if constexpr (U == 2) FRAD_NT_STORE(v, FRAD_GPTR(v2d, dst + u * U) + i);      // (64 bit: a wave's store is one contiguous 1 KiB row)
            else *(FRAD_GPTR(v2d, dst + u * U) + i) = v;
        }
    }
    if (chunk == 0)
        for (long long i = units * U + threadIdx.x; i < NC; i += blockDim.x)
            dst[i] = code_to_f64(code_from_bytes(src, i, BITS, le), BITS);
}

// 16 / 32-bit depths: one thread = two values = one 16-byte output row, so that a wave's store instruction
// writes 1 KiB of contiguous float64 (the unit kernel above stores 16 bytes out of every 64 / 32 per lane);
// four pairs are loaded before the first is converted.
template <int BITS>
__global__ void __launch_bounds__(256) k_p4_unpack_pairs(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                         Geom g, int bpf) {
    static_assert(BITS == 16 || BITS == 32, "two values in one 4- or 8-byte load");
    constexpr int NW = BITS / 16;                             // words per pair
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C, pairs = NC / 2;
    const unsigned char* src = payload + f * g.payload_stride;
    double* dst = out + f * NC;
    const bool le = g.le != 0;
    auto emit = [&](long long p, const uint32_t (&w)[NW]) {
        u64 c0, c1;
        if constexpr (BITS == 16) { c0 = w[0] & 0xffffu; c1 = w[0] >> 16; if (!le) { c0 = bswap16((uint32_t)c0); c1 = bswap16((uint32_t)c1); } }
        else { c0 = w[0]; c1 = w[1]; if (!le) { c0 = bswap32((uint32_t)c0); c1 = bswap32((uint32_t)c1); } }
        v2d v = {code_to_f64(c0, BITS), code_to_f64(c1, BITS)};
        FRAD_NT_STORE(v, FRAD_GPTR(v2d, dst) + p);
    };
    const long long step = (long long)bpf * blockDim.x;
    long long p = (long long)chunk * blockDim.x + threadIdx.x;
    for (; p + 3 * step < pairs; p += 4 * step) {
        uint32_t w[4][NW];
#pragma unroll
        for (int b = 0; b < 4; ++b) load_words<NW>(src + (p + b * step) * (NW * 4), w[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) emit(p + b * step, w[b]);
    }
    for (; p < pairs; p += step) {
        uint32_t w[NW];
        load_words<NW>(src + p * (NW * 4), w);
        emit(p, w);
    }
    if (chunk == 0 && threadIdx.x == 0 && (NC & 1))
        dst[NC - 1] = code_to_f64(code_from_bytes(src, NC - 1, BITS, le), BITS);
}

// 12- and 24-bit depths: one thread = 3 / 6 payload bytes = two values = one 16-byte output row, so that a wave's store instruction
// writes 1 KiB of contiguous float64 (the 48-byte unit kernel leaves a lane's sixteen stores 256 bytes apart at 12 bit, the
// 12-byte kernel below two stores 32 bytes apart at 24 bit).  The bytes come out of the aligned word that holds the first of
// them and, when they reach beyond it, the next one (rows are 16-byte aligned and strided, so an aligned word that holds a
// payload byte lies inside the row's stride); four pairs are loaded before the first is converted.
template <int BITS>
__global__ void __launch_bounds__(256) k_p4_unpack_3b(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                      Geom g, int bpf) {
    static_assert(BITS == 12 || BITS == 24, "3 or 6 bytes per pair");
    constexpr int PB = BITS / 4;                              // payload bytes per pair
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C, pairs = NC / 2;
    const unsigned char* src = payload + f * g.payload_stride;
    double* dst = out + f * NC;
    const bool le = g.le != 0;
    auto load = [&](long long p, uint32_t (&w)[2]) {
        const long long o = PB * p;
        const unsigned char* a = src + (o & ~3LL);
        w[0] = *FRAD_GCPTR(uint32_t, a);
        w[1] = (BITS == 24 || (o & 3) >= 2) ? *FRAD_GCPTR(uint32_t, a + 4) : 0u;
    };
    auto emit = [&](long long p, const uint32_t (&w)[2]) {
        const int sh = 8 * (int)((PB * p) & 3);
        const u64 t = (((u64)w[1] << 32) | w[0]) >> sh;       // the pair's payload bytes, first byte lowest
        u64 c0, c1;
        if constexpr (BITS == 12) {
            const uint32_t b0 = (uint32_t)t & 0xffu, b1 = ((uint32_t)t >> 8) & 0xffu, b2 = ((uint32_t)t >> 16) & 0xffu;
            c0 = (b0 << 4) | (b1 >> 4); c1 = ((b1 & 0xfu) << 8) | b2;
        } else {
            const uint32_t lo = (uint32_t)t & 0xffffffu, hi = (uint32_t)(t >> 24) & 0xffffffu;
            c0 = le ? lo : bswap32(lo) >> 8; c1 = le ? hi : bswap32(hi) >> 8;
        }
        v2d v = {code_to_f64(c0, BITS), code_to_f64(c1, BITS)};
        FRAD_NT_STORE(v, FRAD_GPTR(v2d, dst) + p);
    };
    const long long step = (long long)bpf * blockDim.x;
    long long p = (long long)chunk * blockDim.x + threadIdx.x;
    for (; p + 3 * step < pairs; p += 4 * step) {
        uint32_t w[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) load(p + b * step, w[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) emit(p + b * step, w[b]);
    }
    for (; p < pairs; p += step) {
        uint32_t w[2];
        load(p, w);
        emit(p, w);
    }
    if (chunk == 0 && threadIdx.x == 0 && (NC & 1))
        dst[NC - 1] = code_to_f64(code_from_bytes(src, NC - 1, BITS, le), BITS);
}

// 24 / 48-bit depths: one thread = 12 payload bytes = 4 / 2 values, i.e. 32 / 16 contiguous output bytes per lane
// (the 48-byte unit kernel above leaves 128-byte gaps between a lane's stores); four loads in flight per lane.
template <int BITS>
__global__ void __launch_bounds__(256) k_p4_unpack_12b(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                       Geom g, int bpf) {
    static_assert(BITS == 24 || BITS == 48, "12-byte sub-units");
    constexpr int NB = BITS / 8, V = 12 / NB;
    const long long f = blockIdx.x / bpf;
    const int chunk = blockIdx.x - (int)(f * bpf);
    const long long NC = (long long)g.N * g.C, subs = NC / V;
    const unsigned char* src = payload + f * g.payload_stride;
    double* dst = out + f * NC;
    const bool le = g.le != 0;
    auto emit = [&](long long u, const uint32_t (&w)[3]) {
        double val[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            u64 c = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int s = i * NB + j;
                const u64 b = (w[s >> 2] >> (8 * (s & 3))) & 0xffu;
                c |= b << (le ? 8 * j : 8 * (NB - 1 - j));
            }
            val[i] = code_to_f64(c, BITS);
        }
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            v2d v = {val[2 * i], val[2 * i + 1]};
            if constexpr (V == 2) FRAD_NT_STORE(v, FRAD_GPTR(v2d, dst + u * V) + i);      // (48 bit: contiguous rows)
            else *(FRAD_GPTR(v2d, dst + u * V) + i) = v;
        }
    };
    const long long step = (long long)bpf * blockDim.x;
    long long u = (long long)chunk * blockDim.x + threadIdx.x;
    for (; u + 3 * step < subs; u += 4 * step) {
        uint32_t w[4][3];
#pragma unroll
        for (int b = 0; b < 4; ++b) load_words<3>(src + (u + b * step) * 12, w[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) emit(u + b * step, w[b]);
    }
    for (; u < subs; u += step) {
        uint32_t w[3];
        load_words<3>(src + u * 12, w);
        emit(u, w);
    }
    if (chunk == 0)
        for (long long i = subs * V + threadIdx.x; i < NC; i += blockDim.x)
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
// SH >= 0: swizzled complex layout of the FFT kernels (SH = Plan<LOG2M>::SH); SH < 0: plain reals.
template <typename T, int SH>
__device__ __forceinline__ T& xslot(unsigned char* smem, int cf, int slots, int r) {
    if constexpr (SH >= 0) return real_slot<T, SH>(reinterpret_cast<cx<T>*>(smem) + (long long)cf * slots, r);
    else return reinterpret_cast<T*>(smem)[(long long)cf * slots + r];
}

// Stage-in (encode): interleaved PCM of the block's frames -> T in LDS.  PERMUTE applies Makhoul's
// even/odd permutation (FFT kernels); the direct kernels keep time order.
template <typename T, int LG, int SH, bool PERMUTE>
__device__ FRAD_NOINLINE void stage_in_pcm(const unsigned char* __restrict__ pcm, int smem_off, const Geom& g,
                                             long long f0, int nfl, int slots, bool aligned) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C, NC = N * C;
    const int nv = g.n_valid * C;                     // elements actually present per frame
    if (aligned) {
        constexpr int EPC = 16 >> LG;                 // elements per 16-byte chunk
        const int chunks = (NC + EPC - 1) / EPC;      // NC * itemsize is a multiple of 16 here
        auto fetch = [&](int q, uint32_t (&w)[4]) {
            const int fl = q / chunks, ch = q - fl * chunks;
            const unsigned char* src = pcm + ((frame_base(g, f0 + fl) * C) << LG);
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
        };
        const int total = nfl * chunks, TH = blockDim.x;
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {      // format resolved once, not per element
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            auto place_c = [&](int q, const uint32_t (&w)[4]) {
                const int fl = q / chunks, ch = q - fl * chunks;
                const int e0 = ch * EPC;
                int n = e0 / C, c = e0 - n * C;
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const int e = e0 + i;
                    if (e < NC) {
                        const T v = e < nv ? cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(w, i)) : (T)0;
                        xslot<T, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n) = v;
                    }
                    if (++c == C) { c = 0; ++n; }
                }
            };
            int q = threadIdx.x;
            for (; q + 3 * TH < total; q += 4 * TH) {        // four 16-byte chunks in flight per lane
                uint32_t w[4][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) fetch(q + b * TH, w[b]);
#pragma unroll
                for (int b = 0; b < 4; ++b) place_c(q + b * TH, w[b]);
            }
            for (; q < total; q += TH) {
                uint32_t w[4];
                fetch(q, w);
                place_c(q, w);
            }
        });
    } else {
        for (int q = threadIdx.x; q < nfl * NC; q += blockDim.x) {
            const int fl = q / NC, e = q - fl * NC;
            const int n = e / C, c = e - n * C;
            const unsigned char* src = pcm + ((frame_base(g, f0 + fl) * C) << LG);
            const T v = e < nv ? cvt_pcm<T>(load_raw(src + ((long long)e << LG), LG), g.dtype, g.raw_be) : (T)0;
            xslot<T, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n) = v;
        }
    }
}

// Quad stage-in (FFT kernels): four consecutive sample-frames 4q..4q+3 of one channel are exactly
// the two packed points z[q] = (x[4q], x[4q+2]) and z[M-1-q] = (x[4q+3], x[4q+1]) of Makhoul's
// permutation, so every LDS write is one whole complex slot (16 B in float64), lane-contiguous and
// conflict-free, instead of four scattered 8-byte writes.  Requires 16-byte aligned frames,
// N % 4 == 0 and a row size (C * itemsize) that tiles 16 bytes: mode 1 = row divides 16 (>= 4 rows
// per 16-byte load, LC = log2 C), mode 2 = row of 8 bytes, mode 3 = row a multiple of 16 bytes.
template <typename T, int LG, int SH, int LC>
__device__ __forceinline__ void stage_in_quads_small(const unsigned char* __restrict__ pcm, unsigned char* smem, const Geom& g,
                                                     long long f0, int nfl, int slots) {
    constexpr int EPC = 16 >> LG, C = 1 << LC, ROWS = EPC / C, GPC = ROWS / 4;
    static_assert(GPC >= 1, "row must divide 4 rows into 16 bytes");
    const int M = g.N >> 1;
    const int chunks = (g.N * C) / EPC;
    // format resolved once (dispatch_pcm); four 16-byte chunks in flight per lane
    dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
        constexpr int CODE = decltype(code_tag)::value;
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        auto fetch = [&](int q, uint32_t (&w)[4]) {
            const int fl = q / chunks, ch = q - fl * chunks;
            load_words<4>(pcm + (((f0 + fl) * g.frame_stride * C) << LG) + (long long)ch * 16, w);
        };
        auto place = [&](int q, const uint32_t (&w)[4]) {
            const int fl = q / chunks, ch = q - fl * chunks;
#pragma unroll
            for (int gi = 0; gi < GPC; ++gi) {
                const int zq = ch * GPC + gi;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)(fl * C + c) * slots;
                    T e[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(w, (gi * 4 + i) * C + c));
                    buf[phys<T, SH>(zq)] = cx<T>{e[0], e[2]};
                    buf[phys<T, SH>(M - 1 - zq)] = cx<T>{e[3], e[1]};
                }
            }
        };
        const int total = nfl * chunks, TH = blockDim.x;
        int q = threadIdx.x;
        for (; q + 3 * TH < total; q += 4 * TH) {
            uint32_t w[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) fetch(q + b * TH, w[b]);
#pragma unroll
            for (int b = 0; b < 4; ++b) place(q + b * TH, w[b]);
        }
        for (; q < total; q += TH) {
            uint32_t w[4];
            fetch(q, w);
            place(q, w);
        }
    });
}

template <typename T, int LG, int SH>
__device__ FRAD_NOINLINE void stage_in_quads(const unsigned char* __restrict__ pcm, int smem_off, const Geom& g,
                                               long long f0, int nfl, int slots) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int M = g.N >> 1, C = g.C;
    if (g.in_mode == 1) {
        if constexpr (LG <= 2) {
            const int rb = C << LG;
            if (rb == 1) { if constexpr (LG == 0) stage_in_quads_small<T, LG, SH, 0>(pcm, smem, g, f0, nfl, slots); }
            else if (rb == 2) { if constexpr (LG <= 1) stage_in_quads_small<T, LG, SH, 1 - LG>(pcm, smem, g, f0, nfl, slots); }
            else { stage_in_quads_small<T, LG, SH, 2 - LG>(pcm, smem, g, f0, nfl, slots); }
        }
    } else if (g.in_mode == 2) {                         // row = 8 bytes: C = 8 >> LG, a quad = 32 contiguous bytes
        constexpr int CC = 8 >> LG;
        const int quads = g.N / 4, total = nfl * quads, TH = blockDim.x;
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {       // format resolved once; two quads in flight
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            auto fetch = [&](int q, uint32_t (&w)[8]) {
                const int fl = q / quads, zq = q - fl * quads;
                load_words<8>(pcm + (((f0 + fl) * g.frame_stride * CC) << LG) + (long long)zq * 32, w);
            };
            auto place = [&](int q, const uint32_t (&w)[8]) {
                const int fl = q / quads, zq = q - fl * quads;
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)(fl * CC + c) * slots;
                    T e[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(w, i * CC + c));
                    buf[phys<T, SH>(zq)] = cx<T>{e[0], e[2]};
                    buf[phys<T, SH>(M - 1 - zq)] = cx<T>{e[3], e[1]};
                }
            };
            int q = threadIdx.x;
            for (; q + TH < total; q += 2 * TH) {
                uint32_t w0[8], w1[8];
                fetch(q, w0); fetch(q + TH, w1);
                place(q, w0); place(q + TH, w1);
            }
            for (; q < total; q += TH) { uint32_t w[8]; fetch(q, w); place(q, w); }
        });
    } else {                                             // row a multiple of 16 bytes: (quad, 16-byte column slab)
        constexpr int EPC = 16 >> LG;
        const int slabs = (C << LG) / 16, quads = g.N / 4;
        const long long rowb = (long long)C << LG;
        const int total = nfl * quads * slabs, TH = blockDim.x;
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {       // format resolved once; two tasks (8 loads) in flight
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            auto fetch = [&](int q, uint32_t (&w)[4][4]) {
                const int fl = q / (quads * slabs), r = q - fl * quads * slabs;
                const int zq = r / slabs, sl = r - zq * slabs;
                const unsigned char* src = pcm + (((f0 + fl) * g.frame_stride * C) << LG) + (long long)zq * 4 * rowb + sl * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) load_words<4>(src + i * rowb, w[i]);
            };
            auto place = [&](int q, const uint32_t (&w)[4][4]) {
                const int fl = q / (quads * slabs), r = q - fl * quads * slabs;
                const int zq = r / slabs, sl = r - zq * slabs;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)(fl * C + sl * EPC + e) * slots;
                    T v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(w[i], e));
                    buf[phys<T, SH>(zq)] = cx<T>{v[0], v[2]};
                    buf[phys<T, SH>(M - 1 - zq)] = cx<T>{v[3], v[1]};
                }
            };
            int q = threadIdx.x;
            for (; q + TH < total; q += 2 * TH) {
                uint32_t w0[4][4], w1[4][4];
                fetch(q, w0); fetch(q + TH, w1);
                place(q, w0); place(q + TH, w1);
            }
            for (; q < total; q += TH) { uint32_t w[4][4]; fetch(q, w); place(q, w); }
        });
    }
}

// Pairwise pack-out (FFT kernels, C = CC in {1, 2}): bins k (even) and k+1 of a channel are one
// complex slot, so a thread reads whole slots and emits whole 16-byte payload lines.
template <typename T, int BITS, int SH, int CC>
__device__ __forceinline__ void pack_out_pairs(unsigned char* smem, unsigned char* __restrict__ payload, double* absmax,
                                               const Geom& g, long long f0, int nfl, int slots) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int V = U > 2 * CC ? U : 2 * CC;            // values per thread: whole units and whole slot pairs
    constexpr int KB = V / CC;                            // bins per thread (even)
    const bool le = g.le && (BITS % 8 == 0);
    const int tasks = (g.N * CC) / V;
    for (int fl = 0; fl < nfl; ++fl) {
        unsigned char* dst = payload + (f0 + fl) * g.payload_stride;
        u64 mx = 0;
        for (int u = threadIdx.x; u < tasks; u += blockDim.x) {
            u64 codes[V];
            const int s0 = (u * KB) >> 1;                 // first complex slot
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const cx<T>* buf = reinterpret_cast<const cx<T>*>(smem) + (long long)(fl * CC + c) * slots;
#pragma unroll
                for (int kk = 0; kk < KB / 2; ++kk) {
                    const cx<T> z = buf[phys<T, SH>(s0 + kk)];
                    const u64 a = abs_bits((double)z.x), b = abs_bits((double)z.y);
                    mx = a > mx ? a : mx; mx = b > mx ? b : mx;
                    codes[(2 * kk) * CC + c] = storage_code<T>(z.x, BITS);
                    codes[(2 * kk + 1) * CC + c] = storage_code<T>(z.y, BITS);
                }
            }
#pragma unroll
            for (int w = 0; w < V / U; ++w) {
                u64 unit[U];
#pragma unroll
                for (int i = 0; i < U; ++i) unit[i] = codes[w * U + i];
                uint32_t out[UB / 4];
                pack_unit<BITS>(unit, le, out);
                store_words<UB / 4>(dst + ((long long)u * (V / U) + w) * UB, out);
            }
        }
        block_absmax_commit(mx, absmax, f0 + fl);
    }
}
template <typename T, int SH, int CC>
__device__ FRAD_NOINLINE void pack_out_pairs_any(int smem_off, unsigned char* __restrict__ payload, double* absmax,
                                                   const Geom& g, long long f0, int nfl, int slots) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    switch (g.bits) {
        case 12: pack_out_pairs<T, 12, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
        case 16: pack_out_pairs<T, 16, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
        case 24: pack_out_pairs<T, 24, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
        case 32: pack_out_pairs<T, 32, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
        case 48: pack_out_pairs<T, 48, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
        default: pack_out_pairs<T, 64, SH, CC>(smem, payload, absmax, g, f0, nfl, slots); break;
    }
}

// Pairwise unpack-in (decode mirror of pack_out_pairs).
template <int BITS, int SH, int CC>
__device__ __forceinline__ void unpack_in_pairs(const unsigned char* __restrict__ payload, unsigned char* smem, const Geom& g,
                                                long long f0, int nfl, int slots) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int V = U > 2 * CC ? U : 2 * CC, KB = V / CC;
    const bool le = g.le && (BITS % 8 == 0);
    const int tasks = (g.N * CC) / V;
    constexpr int WPT = (V / U) * (UB / 4);                   // words per task
    auto fetch = [&](int q, uint32_t (&in)[WPT]) {
        const int fl = q / tasks, u = q - fl * tasks;
        const unsigned char* src = payload + (f0 + fl) * g.payload_stride;
        load_words<WPT>(src + (long long)u * (V / U) * UB, in);
    };
    auto place = [&](int q, const uint32_t (&in)[WPT]) {
        const int fl = q / tasks, u = q - fl * tasks;
        u64 codes[V];
#pragma unroll
        for (int w = 0; w < V / U; ++w) {
            uint32_t part[UB / 4];
#pragma unroll
            for (int i = 0; i < UB / 4; ++i) part[i] = in[w * (UB / 4) + i];
            u64 unit[U];
            unpack_unit<BITS>(part, le, unit);
#pragma unroll
            for (int i = 0; i < U; ++i) codes[w * U + i] = unit[i];
        }
        const int s0 = (u * KB) >> 1;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)(fl * CC + c) * slots;
#pragma unroll
            for (int kk = 0; kk < KB / 2; ++kk)
                buf[phys<double, SH>(s0 + kk)] = cx<double>{code_to_f64(codes[(2 * kk) * CC + c], BITS),
                                                            code_to_f64(codes[(2 * kk + 1) * CC + c], BITS)};
        }
    };
    const int total = nfl * tasks, TH = blockDim.x;
    int q = threadIdx.x;
    if constexpr (WPT <= 8) {                                 // up to 32 bytes per task: four tasks in flight per lane
        for (; q + 3 * TH < total; q += 4 * TH) {
            uint32_t in[4][WPT];
#pragma unroll
            for (int b = 0; b < 4; ++b) fetch(q + b * TH, in[b]);
#pragma unroll
            for (int b = 0; b < 4; ++b) place(q + b * TH, in[b]);
        }
    }
    for (; q < total; q += TH) {
        uint32_t in[WPT];
        fetch(q, in);
        place(q, in);
    }
}
template <int SH, int CC>
__device__ FRAD_NOINLINE void unpack_in_pairs_any(const unsigned char* __restrict__ payload, int smem_off, const Geom& g,
                                                    long long f0, int nfl, int slots) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    switch (g.bits) {
        case 12: unpack_in_pairs<12, SH, CC>(payload, smem, g, f0, nfl, slots); break;
        case 16: unpack_in_pairs<16, SH, CC>(payload, smem, g, f0, nfl, slots); break;
        case 24: unpack_in_pairs<24, SH, CC>(payload, smem, g, f0, nfl, slots); break;
        case 32: unpack_in_pairs<32, SH, CC>(payload, smem, g, f0, nfl, slots); break;
        case 48: unpack_in_pairs<48, SH, CC>(payload, smem, g, f0, nfl, slots); break;
        default: unpack_in_pairs<64, SH, CC>(payload, smem, g, f0, nfl, slots); break;
    }
}

// Quad store (decode epilogue, C = CC in {1, 2}): z[q] and z[M-1-q] of a channel are the four
// consecutive samples 4q..4q+3 -> 4*CC contiguous float64 of the interleaved output.
template <int SH, int CC>
__device__ FRAD_NOINLINE void store_pcm_quads(int smem_off, double* __restrict__ out, const Geom& g,
                                                long long f0, int nfl, int slots) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int M = g.N >> 1, quads = g.N / 4;
    for (int q = threadIdx.x; q < nfl * quads; q += blockDim.x) {
        const int fl = q / quads, zq = q - fl * quads;
        double row[4][CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const cx<double>* buf = reinterpret_cast<const cx<double>*>(smem) + (long long)(fl * CC + c) * slots;
            const cx<double> a = buf[phys<double, SH>(zq)], b = buf[phys<double, SH>(M - 1 - zq)];
            row[0][c] = a.x; row[2][c] = a.y; row[3][c] = b.x; row[1][c] = b.y;
        }
        auto dst = FRAD_GPTR(v2d, out + (f0 + fl) * (long long)g.N * CC + (long long)zq * 4 * CC);
        if constexpr (CC == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { v2d v = {row[i][0], row[i][1]}; dst[i] = v; }
        } else {
            v2d v0 = {row[0][0], row[1][0]}, v1 = {row[2][0], row[3][0]};
            dst[0] = v0; dst[1] = v1;
        }
    }
}

// Epilogue (encode): X[k] of every channel-frame in LDS -> absmax, storage cast, pack, store.
// Payload order is bin-major / channel-minor (profile0.py:30: freqs.T.ravel()).
template <typename T, int BITS, int SH>
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
                const T v = xslot<T, SH>(smem, fl * C + c, slots, k);
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
            auto value = [&](long long i) -> T { const int k = (int)(i / C); return xslot<T, SH>(smem, fl * C + (int)(i - (long long)k * C), slots, k); };
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

template <typename T, int SH>
__device__ FRAD_NOINLINE void pack_out_any(int smem_off, unsigned char* __restrict__ payload, double* absmax,
                                             const Geom& g, long long f0, int nfl, int slots, bool aligned) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    switch (g.bits) {
        case 12: pack_out<T, 12, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 16: pack_out<T, 16, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 24: pack_out<T, 24, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 32: pack_out<T, 32, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        case 48: pack_out<T, 48, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
        default: pack_out<T, 64, SH>(smem, payload, absmax, g, f0, nfl, slots, aligned); break;
    }
}

// Stage-in (decode): payload -> unpack -> scrub -> X[k] (float64) in LDS.
template <int BITS, int SH>
__device__ __forceinline__ void unpack_in(const unsigned char* __restrict__ payload, unsigned char* smem, const Geom& g,
                                          long long f0, int nfl, int slots, bool aligned) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    const int C = g.C, NC = g.N * C;
    const bool le = g.le && (BITS % 8 == 0);
    const int units = aligned ? NC / U : 0;
    for (int fl = 0; fl < nfl; ++fl) {
        const unsigned char* src = payload + (f0 + fl) * g.payload_stride;
        auto emit = [&](int u, const uint32_t (&w)[UB / 4]) {
            u64 codes[U];
            unpack_unit<BITS>(w, le, codes);
            int k = (u * U) / C, c = (u * U) - k * C;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                xslot<double, SH>(smem, fl * C + c, slots, k) = code_to_f64(codes[i], BITS);
                if (++c == C) { c = 0; ++k; }
            }
        };
        const int TH = blockDim.x;
        int u = threadIdx.x;
        for (; u + 3 * TH < units; u += 4 * TH) {            // four units in flight per lane before the first conversion
            uint32_t w[4][UB / 4];
#pragma unroll
            for (int b = 0; b < 4; ++b) load_words<UB / 4>(src + (long long)(u + b * TH) * UB, w[b]);
#pragma unroll
            for (int b = 0; b < 4; ++b) emit(u + b * TH, w[b]);
        }
        for (; u < units; u += TH) {
            uint32_t w[UB / 4];
            load_words<UB / 4>(src + (long long)u * UB, w);
            emit(u, w);
        }
        for (int i = units * U + threadIdx.x; i < NC; i += blockDim.x) {
            const int k = i / C, c = i - k * C;
            xslot<double, SH>(smem, fl * C + c, slots, k) = code_to_f64(code_from_bytes(src, i, BITS, le), BITS);
        }
    }
}
template <int SH>
__device__ FRAD_NOINLINE void unpack_in_any(const unsigned char* __restrict__ payload, int smem_off, const Geom& g,
                                              long long f0, int nfl, int slots, bool aligned) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    switch (g.bits) {
        case 12: unpack_in<12, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 16: unpack_in<16, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 24: unpack_in<24, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 32: unpack_in<32, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
        case 48: unpack_in<48, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
        default: unpack_in<64, SH>(payload, smem, g, f0, nfl, slots, aligned); break;
    }
}

// Epilogue (decode): time samples in LDS -> interleaved float64 [N, C] rows, 16 bytes per lane.
// Decode with the caller's output conversion applied in the store (frad_p0_digital_pcm / frad_p1_digital_pcm, backend/pcmformat.py:
// 49-62 + src/decoder.py:23): Geom::dtype names the PCM format the decoded samples leave in (FRAD_PCM_F64LE = 22: plain float64).
// `fn(KIND, LGS)` runs with the format's kind and log2 size as compile-time tags.
template <typename F> __device__ __forceinline__ void dispatch_out_format(int dtype, F&& fn) {
    const int kind = dtype >> 3, lg = (dtype >> 1) & 3;
#define FRAD_OF(K, L) fn(std::integral_constant<int, K>{}, std::integral_constant<int, L>{})
    switch (kind * 4 + lg) {
        case 0: FRAD_OF(0, 0); break; case 1: FRAD_OF(0, 1); break; case 2: FRAD_OF(0, 2); break; case 3: FRAD_OF(0, 3); break;
        case 4: FRAD_OF(1, 0); break; case 5: FRAD_OF(1, 1); break; case 6: FRAD_OF(1, 2); break; case 7: FRAD_OF(1, 3); break;
        case 9: FRAD_OF(2, 1); break; case 10: FRAD_OF(2, 2); break; default: FRAD_OF(2, 3); break;
    }
#undef FRAD_OF
}
template <int LGS> __device__ __forceinline__ void store_pcm_elem(unsigned char* p, u64 b, bool be) {
    if constexpr (LGS == 0) *FRAD_GPTR(unsigned char, p) = (unsigned char)b;
    else if constexpr (LGS == 1) *FRAD_GPTR(unsigned short, p) = (unsigned short)(be ? bswap16((uint32_t)b) : (uint32_t)b);
    else if constexpr (LGS == 2) *FRAD_GPTR(uint32_t, p) = be ? bswap32((uint32_t)b) : (uint32_t)b;
    else *FRAD_GPTR(u64, p) = be ? bswap64(b) : b;
}

template <int SH, bool PERMUTE>
__device__ FRAD_NOINLINE void store_pcm_f64(int smem_off, double* __restrict__ out, const Geom& g,
                                              long long f0, int nfl, int slots) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C, NC = N * C;
    if (g.dtype != 22) {                            // a narrower PCM format: convert on the way out, element by element
        dispatch_out_format(g.dtype, [&](auto kind_tag, auto lg_tag) {
            constexpr int KIND = decltype(kind_tag)::value, LGS = decltype(lg_tag)::value;
            const bool be = (g.dtype & 1) != 0, raw = g.raw_be != 0 && be;
            for (int fl = 0; fl < nfl; ++fl) {
                unsigned char* dst = reinterpret_cast<unsigned char*>(out) + ((frame_base(g, f0 + fl) * C) << LGS);
                for (int e = threadIdx.x; e < NC; e += blockDim.x) {
                    const int n = e / C, c = e - n * C;
                    const double v = xslot<double, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
                    store_pcm_elem<LGS>(dst + ((long long)e << LGS), from_f64_bits<KIND, LGS>(v, raw), be);
                }
            }
        });
        return;
    }
    const int pairs = NC / 2;                       // out + f*NC is 16-byte aligned when NC is even
    for (int fl = 0; fl < nfl; ++fl) {
        double* dst = out + frame_base(g, f0 + fl) * C;       // (decode: frame_stride = N)
        if ((NC & 1) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            for (int p = threadIdx.x; p < pairs; p += blockDim.x) {
                const int e = 2 * p;
                int n = e / C, c = e - n * C;
                const double v0 = xslot<double, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
                if (++c == C) { c = 0; ++n; }
                const double v1 = xslot<double, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
                v2d v = {v0, v1};
                *(FRAD_GPTR(v2d, dst) + p) = v;
            }
        } else {
            for (int e = threadIdx.x; e < NC; e += blockDim.x) {
                const int n = e / C, c = e - n * C;
                dst[e] = xslot<double, SH>(smem, fl * C + c, slots, PERMUTE ? makhoul(n, N) : n);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Channel-group mode: a frame whose channels do not fit the 160 KiB LDS together (e.g. 7.1 at
// N = 4096 in float64) is transformed `cg` channels at a time by one block.  The interleaved
// payload / PCM rows are then touched per value instead of per 16-byte unit.
// ---------------------------------------------------------------------------------------------
template <typename T, int LG, int SH>
__device__ FRAD_NOINLINE void stage_in_pcm_group(const unsigned char* __restrict__ pcm, int smem_off, const Geom& g,
                                                   long long f, int slots, int c0, int cgn) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C;
    const unsigned char* src = pcm + ((frame_base(g, f) * C) << LG);
    const int total = N * cgn, TH = blockDim.x;
    int q0 = threadIdx.x;
    if (g.n_valid == N) {                                    // batches of 8 element loads in flight per lane, format resolved once
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto code_tag, auto raw_tag) {
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            for (; q0 + 7 * TH < total; q0 += 8 * TH) {
                u64 raw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i * TH, n = q / cgn, j = q - n * cgn;
                    raw[i] = load_raw(src + (((long long)n * C + c0 + j) << LG), LG);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i * TH, n = q / cgn, j = q - n * cgn;
                    xslot<T, SH>(smem, j, slots, makhoul(n, N)) = cvt_pcm_c<T, CODE, RAW>(raw[i]);
                }
            }
        });
    }
    for (int q = q0; q < total; q += TH) {
        const int n = q / cgn, j = q - n * cgn;
        const long long e = (long long)n * C + c0 + j;
        const T v = n < g.n_valid ? cvt_pcm<T>(load_raw(src + (e << LG), LG), g.dtype, g.raw_be) : (T)0;
        xslot<T, SH>(smem, j, slots, makhoul(n, N)) = v;
    }
}

template <typename T, int SH>
__device__ FRAD_NOINLINE void pack_out_group(int smem_off, unsigned char* __restrict__ payload, double* absmax,
                                               const Geom& g, long long f, int slots, int c0, int cgn) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C, bits = g.bits;
    const bool le = g.le && (bits % 8 == 0);
    unsigned char* dst = payload + f * g.payload_stride;
    u64 mx = 0;
    if (bits == 12) {                                   // host guarantees C, c0 and cgn even: pairs stay in the group
        const int half = cgn / 2;
        for (int q = threadIdx.x; q < N * half; q += blockDim.x) {
            const int k = q / half, p = q - k * half;
            const T a = xslot<T, SH>(smem, 2 * p, slots, k), b = xslot<T, SH>(smem, 2 * p + 1, slots, k);
            const u64 ma = abs_bits((double)a), mb = abs_bits((double)b);
            mx = ma > mx ? ma : mx; mx = mb > mx ? mb : mx;
            const uint32_t ca = (uint32_t)storage_code<T>(a, 12), cb = (uint32_t)storage_code<T>(b, 12);
            const long long o = ((long long)k * C + c0 + 2 * p) / 2 * 3;
            dst[o] = (unsigned char)(ca >> 4); dst[o + 1] = (unsigned char)(((ca & 15) << 4) | (cb >> 8)); dst[o + 2] = (unsigned char)cb;
        }
    } else {
        const int nb = bits >> 3;
        const bool sized = (nb == 2 || nb == 4 || nb == 8) && (reinterpret_cast<uintptr_t>(dst) % nb == 0);
        for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {
            const int k = q / cgn, j = q - k * cgn;
            const T v = xslot<T, SH>(smem, j, slots, k);
            const u64 a = abs_bits((double)v);
            mx = a > mx ? a : mx;
            const u64 code = storage_code<T>(v, bits);
            const long long o = ((long long)k * C + c0 + j) * nb;
            if (sized) {                                      // one element-sized store
                if (nb == 4) *FRAD_GPTR(uint32_t, dst + o) = le ? (uint32_t)code : bswap32((uint32_t)code);
                else if (nb == 8) *FRAD_GPTR(u64, dst + o) = le ? code : bswap64(code);
                else *FRAD_GPTR(unsigned short, dst + o) = (unsigned short)(le ? (uint32_t)code : bswap16((uint32_t)code));
            } else {
                for (int b = 0; b < nb; ++b) dst[o + b] = (unsigned char)code_byte(code, bits, le, b);
            }
        }
    }
    block_absmax_commit(mx, absmax, f);
}

template <int SH>
__device__ FRAD_NOINLINE void unpack_in_group(const unsigned char* __restrict__ payload, int smem_off, const Geom& g,
                                                long long f, int slots, int c0, int cgn) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C;
    const bool le = g.le && (g.bits % 8 == 0);
    const unsigned char* src = payload + f * g.payload_stride;
    const bool sized = (g.bits == 16 || g.bits == 32 || g.bits == 64) && (reinterpret_cast<uintptr_t>(src) % (g.bits / 8) == 0);
    const int total = N * cgn, T = blockDim.x;
    int q0 = threadIdx.x;
    {
        // The group's share of a payload row (cgn values) is contiguous: when it is whole, aligned 16-byte pieces, a
        // lane loads a piece at a time, 8 pieces in flight -- one block per CU runs here, so only bytes in flight
        // per lane hide the memory latency (element loads reach ~0.6 TB/s).  64-bit values already move 8 bytes per
        // element load and measured faster through the batched path below.
        const int nbv = g.bits >> 3;                          // bytes per value (16/32/64 bit)
        const long long rowb = (long long)C * nbv, pieceb = (long long)cgn * nbv;
        if (sized && nbv <= 4 && pieceb % 16 == 0 && rowb % 16 == 0 && ((long long)c0 * nbv) % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const int ppr = (int)(pieceb / 16), vpp = 16 / nbv;          // pieces per row, values per piece
            const int pieces = N * ppr;
            int p0 = threadIdx.x;
            auto emit = [&](int p, const uint32_t (&w)[4]) {
                const int k = p / ppr, pi = p - k * ppr;
                for (int v = 0; v < vpp; ++v) {
                    u64 c;
                    if (nbv == 2) { c = (w[v >> 1] >> (16 * (v & 1))) & 0xffffu; if (!le) c = bswap16((uint32_t)c); }
                    else if (nbv == 4) { c = w[v]; if (!le) c = bswap32((uint32_t)c); }
                    else { c = (u64)w[2 * v] | ((u64)w[2 * v + 1] << 32); if (!le) c = bswap64(c); }
                    xslot<double, SH>(smem, pi * vpp + v, slots, k) = code_to_f64(c, g.bits);
                }
            };
            for (; p0 + 7 * T < pieces; p0 += 8 * T) {
                uint32_t w[8][4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p = p0 + i * T, k = p / ppr, pi = p - k * ppr;
                    load_words<4>(src + (long long)k * rowb + (long long)c0 * nbv + pi * 16, w[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) emit(p0 + i * T, w[i]);
            }
            for (int p = p0; p < pieces; p += T) {
                const int k = p / ppr, pi = p - k * ppr;
                uint32_t w[4];
                load_words<4>(src + (long long)k * rowb + (long long)c0 * nbv + pi * 16, w);
                emit(p, w);
            }
            return;
        }
    }
    if (sized) {
        // batches of 8 element loads in flight per lane
        const int lgb = g.bits == 16 ? 1 : g.bits == 32 ? 2 : 3;
        for (; q0 + 7 * T < total; q0 += 8 * T) {
            u64 code[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = q0 + i * T, k = q / cgn, j = q - k * cgn;
                code[i] = load_raw(src + (((long long)k * C + c0 + j) << lgb), lgb);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = q0 + i * T, k = q / cgn, j = q - k * cgn;
                u64 c = code[i];
                if (!le) c = lgb == 1 ? bswap16((uint32_t)c) : lgb == 2 ? bswap32((uint32_t)c) : bswap64(c);
                xslot<double, SH>(smem, j, slots, k) = code_to_f64(c, g.bits);
            }
        }
    }
    for (int q = q0; q < total; q += T) {
        const int k = q / cgn, j = q - k * cgn;
        const long long idx = (long long)k * C + c0 + j;
        u64 code;
        if (sized) {                                          // one element-sized load instead of byte loads
            const int lgb = g.bits == 16 ? 1 : g.bits == 32 ? 2 : 3;
            code = load_raw(src + (idx << lgb), lgb);
            if (!le) code = lgb == 1 ? bswap16((uint32_t)code) : lgb == 2 ? bswap32((uint32_t)code) : bswap64(code);
        } else {
            code = code_from_bytes(src, idx, g.bits, le);
        }
        xslot<double, SH>(smem, j, slots, k) = code_to_f64(code, g.bits);
    }
}

template <int SH>
__device__ FRAD_NOINLINE void store_pcm_group(int smem_off, double* __restrict__ out, const Geom& g,
                                                long long f, int slots, int c0, int cgn) {
    FRAD_DYN_SMEM(smem_base_);
    unsigned char* smem = smem_base_ + smem_off;
    const int N = g.N, C = g.C;
    if (g.dtype != 22) {                            // (see store_pcm_f64)
        dispatch_out_format(g.dtype, [&](auto kind_tag, auto lg_tag) {
            constexpr int KIND = decltype(kind_tag)::value, LGS = decltype(lg_tag)::value;
            const bool be = (g.dtype & 1) != 0, raw = g.raw_be != 0 && be;
            unsigned char* dst = reinterpret_cast<unsigned char*>(out) + ((f * (long long)N * C) << LGS);
            for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {
                const int n = q / cgn, j = q - n * cgn;
                const double v = xslot<double, SH>(smem, j, slots, makhoul(n, N));
                store_pcm_elem<LGS>(dst + (((long long)n * C + c0 + j) << LGS), from_f64_bits<KIND, LGS>(v, raw), be);
            }
        });
        return;
    }
    double* dst = out + f * (long long)N * C;
    if ((cgn & 1) == 0 && (C & 1) == 0 && (c0 & 1) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int half = cgn / 2;                             // two channels = one 16-byte store
        for (int q = threadIdx.x; q < N * half; q += blockDim.x) {
            const int n = q / half, j = (q - n * half) * 2, m = makhoul(n, N);
            v2d v = {xslot<double, SH>(smem, j, slots, m), xslot<double, SH>(smem, j + 1, slots, m)};
            *FRAD_GPTR(v2d, dst + (long long)n * C + c0 + j) = v;
        }
        return;
    }
    for (int q = threadIdx.x; q < N * cgn; q += blockDim.x) {
        const int n = q / cgn, j = q - n * cgn;
        dst[(long long)n * C + c0 + j] = xslot<double, SH>(smem, j, slots, makhoul(n, N));
    }
}

// =============================================================================================
// K3/K5  profile 0 encode, N = 2^(LOG2M+1): LDS-resident FFT.
// block = fpb frames x C channels x TEAM lanes (rounded up to whole waves).
// =============================================================================================
template <typename T, int LOG2M, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 2 : 1)) k_p0_fwd(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                  double* absmax, const cx<T>* __restrict__ tw, const cx<T>* __restrict__ post,
                                                  Geom g, int aligned_in, int aligned_out) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    // blockDim.x == fpb * C * TEAM exactly (a whole number of waves): every team owns a buffer.
    // In a short last block the spare teams transform whatever the LDS holds and nobody reads
    // their result; that keeps the multi-wave teams' barriers uniform.
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)cf * SLOTS;
    if (g.in_mode) stage_in_quads<T, LG, SH>(pcm, 0, g, f0, nfl, SLOTS);
    else stage_in_pcm<T, LG, SH, true>(pcm, 0, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    fft_team<T, LOG2M, false>(buf, t, tw);
    dct_post<T, LOG2M>(buf, t, post);
    __syncthreads();
    if (g.cc_fast == 2) pack_out_pairs_any<T, SH, 2>(0, payload, absmax, g, f0, nfl, SLOTS);
    else if (g.cc_fast == 1) pack_out_pairs_any<T, SH, 1>(0, payload, absmax, g, f0, nfl, SLOTS);
    else pack_out_any<T, SH>(0, payload, absmax, g, f0, nfl, SLOTS, aligned_out != 0);
}

// Channel-group variant: one frame per block, `g.cg` channels per pass (see above).  A kernel of
// its own so that the common kernel's register allocation is not disturbed by the second copy of
// the transform.
template <typename T, int LOG2M, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_p0_fwd_grp(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                      double* absmax, const cx<T>* __restrict__ tw, const cx<T>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<T>* buf = reinterpret_cast<cx<T>*>(smem) + (long long)cf * SLOTS;
    for (int c0 = 0; c0 < g.C; c0 += g.cg) {
        const int cgn = g.C - c0 < g.cg ? g.C - c0 : g.cg;
        stage_in_pcm_group<T, LG, SH>(pcm, 0, g, f0, SLOTS, c0, cgn);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);                          // keep per-lane LDS addresses out of the loop preheader
        fft_team<T, LOG2M, false>(buf, tt, tw);
        dct_post<T, LOG2M>(buf, tt, post);
        __syncthreads();
        pack_out_group<T, SH>(0, payload, absmax, g, f0, SLOTS, c0, cgn);
        __syncthreads();
    }
}

// =============================================================================================
// K4  profile 0 decode (always float64, as the reference widens before idct).
// =============================================================================================
template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 2 : 1)) k_p0_inv(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                  const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post,
                                                  Geom g, int aligned_in) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    if (g.cc_fast == 2) unpack_in_pairs_any<SH, 2>(payload, 0, g, f0, nfl, SLOTS);
    else if (g.cc_fast == 1) unpack_in_pairs_any<SH, 1>(payload, 0, g, f0, nfl, SLOTS);
    else unpack_in_any<SH>(payload, 0, g, f0, nfl, SLOTS, aligned_in != 0);
    __syncthreads();
    dct_pre_inverse<double, LOG2M>(buf, t, post);
    fft_team<double, LOG2M, true>(buf, t, tw);
    __syncthreads();
    if (g.in_mode == 2) store_pcm_quads<SH, 2>(0, out, g, f0, nfl, SLOTS);
    else if (g.in_mode == 1) store_pcm_quads<SH, 1>(0, out, g, f0, nfl, SLOTS);
    else store_pcm_f64<SH, true>(0, out, g, f0, nfl, SLOTS);
}

template <int LOG2M, int MAXT>
__global__ void __launch_bounds__(MAXT) k_p0_inv_grp(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                      const cx<double>* __restrict__ tw, const cx<double>* __restrict__ post, Geom g) {
    constexpr int M = 1 << LOG2M, TEAM = Plan<LOG2M>::TEAM, SLOTS = padded_slots(M), SH = Plan<LOG2M>::SH;
    FRAD_DYN_SMEM(smem);
    const long long f0 = blockIdx.x;
    const int cf = threadIdx.x / TEAM, t = threadIdx.x - cf * TEAM;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem) + (long long)cf * SLOTS;
    for (int c0 = 0; c0 < g.C; c0 += g.cg) {
        const int cgn = g.C - c0 < g.cg ? g.C - c0 : g.cg;
        unpack_in_group<SH>(payload, 0, g, f0, SLOTS, c0, cgn);
        __syncthreads();
        int tt = t; FRAD_OPAQUE(tt);                          // keep per-lane LDS addresses out of the loop preheader
        dct_pre_inverse<double, LOG2M>(buf, tt, post);
        fft_team<double, LOG2M, true>(buf, tt, tw);
        __syncthreads();
        store_pcm_group<SH>(0, out, g, f0, SLOTS, c0, cgn);
        __syncthreads();
    }
}

// absmax[i] > lim for any i -> *flag |= 1 (profile0.py:24-26 as one batch test; NaN compares false)
template <int DUMMY>
__global__ void __launch_bounds__(1024) k_overflow_scan(const double* __restrict__ absmax, long long n, double lim, int* flag) {
    bool over = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        over |= absmax[i] > lim;
    if (over) atomicOr(flag, 1);
}

// =============================================================================================
// direct kernels: any N.  cos table ct[j] = cos(pi * j / (2N)), j in [0, 4N).
// LDS: x (T) and X (T) per channel-frame, plain arrays of N reals each.
// =============================================================================================
// Both are a dense cosine product, so the work is tiled like one: a thread owns one output index and up
// to 8 channel-frames (columns) at a time, so a table value fetched for (k, n) feeds 8 FMAs and the LDS
// operand reads are wave-wide broadcasts.  g.fpb frames per block (host: 8 / C, fewer when that would leave CUs idle).
template <typename T, bool FWD>
__device__ __forceinline__ void direct_product(const T* __restrict__ src, T* __restrict__ dst, const double* __restrict__ ct,
                                               int N, int cols) {
    const unsigned fourN = 4u * (unsigned)N;
    const double inv_n = 1.0 / (double)N;
    for (int o = threadIdx.x; o < N; o += blockDim.x) {
        // forward: o = k, table index k (2n + 1) mod 4N over n;  inverse: o = n, same index over k >= 1
        const unsigned step = FWD ? (2u * (unsigned)o) % fourN : (2u * (unsigned)o + 1u) % fourN;
        for (int c0 = 0; c0 < cols; c0 += 8) {
            const int nb = cols - c0 < 8 ? cols - c0 : 8;
            const T* sb = src + (long long)c0 * N;
            double acc[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b] = 0.0;
            unsigned jj = FWD ? (unsigned)o % fourN : step;
            if (nb == 8) {
#pragma unroll 4
                for (int i = FWD ? 0 : 1; i < N; ++i) {
                    const double w = ct[jj];
#pragma unroll
                    for (int b = 0; b < 8; ++b) acc[b] = fma((double)sb[(long long)b * N + i], w, acc[b]);
                    jj += step; if (jj >= fourN) jj -= fourN;
                }
            } else {
#pragma unroll 2
                for (int i = FWD ? 0 : 1; i < N; ++i) {
                    const double w = ct[jj];
#pragma unroll
                    for (int b = 0; b < 8; ++b) if (b < nb) acc[b] = fma((double)sb[(long long)b * N + i], w, acc[b]);
                    jj += step; if (jj >= fourN) jj -= fourN;
                }
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) if (b < nb) {
                if constexpr (FWD) dst[(long long)(c0 + b) * N + o] = (T)(acc[b] * inv_n);
                else dst[(long long)(c0 + b) * N + o] = (T)((double)sb[(long long)b * N] + 2.0 * acc[b]);
            }
        }
    }
}

template <typename T, int LG>
__global__ void __launch_bounds__(1024) k_p0_fwd_direct(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload,
                                                        double* absmax, const double* __restrict__ ct, Geom g,
                                                        int aligned_in, int aligned_out) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    T* x = reinterpret_cast<T*>(smem);
    T* X = x + (long long)N * C * g.fpb;
    stage_in_pcm<T, LG, -1, false>(pcm, 0, g, f0, nfl, N, aligned_in != 0);
    __syncthreads();
    direct_product<T, true>(x, X, ct, N, nfl * C);
    __syncthreads();
    pack_out_any<T, -1>((int)((long long)N * C * g.fpb * sizeof(T)), payload, absmax, g, f0, nfl, N, aligned_out != 0);
}

template <int UNUSED>
__global__ void __launch_bounds__(1024) k_p0_inv_direct(const unsigned char* __restrict__ payload, double* __restrict__ out,
                                                        const double* __restrict__ ct, Geom g, int aligned_in) {
    FRAD_DYN_SMEM(smem);
    const int N = g.N, C = g.C;
    const long long f0 = (long long)blockIdx.x * g.fpb;
    const long long rem = g.n_frames - f0;
    const int nfl = rem < g.fpb ? (int)rem : g.fpb;
    double* X = reinterpret_cast<double*>(smem);
    double* x = X + (long long)N * C * g.fpb;
    unpack_in_any<-1>(payload, 0, g, f0, nfl, N, aligned_in != 0);
    __syncthreads();
    direct_product<double, false>(X, x, ct, N, nfl * C);
    __syncthreads();
    store_pcm_f64<-1, false>((int)((long long)N * C * g.fpb * sizeof(double)), out, g, f0, nfl, N);
}

}  // namespace frad
