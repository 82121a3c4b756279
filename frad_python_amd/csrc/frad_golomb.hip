// frad_golomb.hip -- profile 1's Exp-Golomb-Rice stage on the device.
//
// Reference: fourier/tools/p1tools.py:46-60 (exp_golomb_rice_encode), :62-74 (exp_golomb_rice_decode) and the frame
// body layout of fourier/profile1.py:43-45 / :59-64:
//
//     body = struct.pack('>I', len(thres_gol)) + thres_gol + freqs_gol        (then deflated on the host)
//     *_gol = byte k, then per value v the zig-zag code z (v > 0: 2v - 1, else -2v) written as m zeros followed by
//             the (m + k + 1)-bit binary of z + 2^k, MSB first, zero-padded to a byte;  k = ceil(log2(max |v|))
//
// With k taken from the maximum, z + 2^k < 2^(k+2): every value costs k + 1 or k + 3 bits, so a frame's worst case is
// known up front (frad_p1_golomb_bound) and the coder needs no second look at the data.
//
//   k_gol_encode   one WAVE per frame: |max| reduction -> k, per-lane code lengths of 16 consecutive values -> wave scan
//                  -> bit offsets -> every lane packs its stretch into whole words and ORs them into an LDS word buffer ->
//                  big-endian words out.  A tile is 1024 values; the partial last word is carried into the next tile.
//   k_rows_scan / k_rows_gather   exclusive scan of the body lengths and a gather into one contiguous buffer, so
//                  that a batch goes back to the host (zlib) as ONE copy of exactly the bytes it needs.
//   k_gol_decode   one LANE per frame (thresholds, then coefficients): the code boundaries of a prefix code are a
//                  sequential dependency, so the parallelism is across frames; the wave stages 256 bytes of every
//                  lane's stream through LDS per round.  Semantics of the reference decoder incl. its end conditions (a
//                  run of zero bits ends the stream; a code cut short by the end of the buffer is read from the bits there).
#include "frad_common.hpp"
#include "../../include/frad_hip.h"
#include <cstdlib>

namespace frad {
namespace {

constexpr int GT = 256;                    // threads of the row scan / gather blocks
constexpr int GV = 16;                     // consecutive values per lane and tile
constexpr int WTILE = 64 * GV;             // values per tile of the coder (one wave)
constexpr int GWORDS = (WTILE * 35 + 31) / 32 + 4;     // 32-bit words a tile can fill (<= 35 bits per value) + carry
constexpr int GOL_LDS = GWORDS * 4;

__device__ __forceinline__ int bitlen64(u64 v) { return v ? 64 - __builtin_clzll(v) : 0; }
__device__ __forceinline__ u64 zigzag(int32_t v) { return v > 0 ? 2ull * (u64)v - 1ull : 2ull * (u64)(-(long long)v); }
// k = ceil(log2(dmax)) for dmax >= 1 (numpy evaluates it in float64, exact for |v| < 2^53), 0 for dmax = 0
__device__ __forceinline__ int rice_k(u64 dmax) { return dmax <= 1 ? 0 : bitlen64(dmax - 1); }

// block-wide inclusive scan of one int per thread (GT threads); returns this thread's inclusive sum, *total = block sum
__device__ __forceinline__ long long block_scan(long long v, long long* tmp, long long* total) {
    tmp[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < GT; off <<= 1) {
        const long long add = threadIdx.x >= (unsigned)off ? tmp[threadIdx.x - off] : 0;
        __syncthreads();
        tmp[threadIdx.x] += add;
        __syncthreads();
    }
    const long long r = tmp[threadIdx.x];
    *total = tmp[GT - 1];
    __syncthreads();
    return r;
}
// ---- encode, one WAVE per frame ------------------------------------------------------------------------------------------
// The body is ONE bit sequence.  A tile is 64 lanes x 16 consecutive values; a lane's codes are contiguous in the stream, so
// it packs them into whole 32-bit words in registers (MSB first) and ORs only those into the LDS word buffer -- the first and
// the last word of its stretch are shared with its neighbours.  Bit offsets: a wave scan of the lanes' code lengths (no block
// barrier anywhere).  After a tile the complete words leave as big-endian words (coalesced) and are zeroed in the same
// sweep; the partial last word stays as word 0 of the next tile.
struct WaveSink {
    int* words;                            // LDS, bits [base, base + 32 * GWORDS), MSB-first inside a word; all zero between tiles
    unsigned char* out;                    // the frame's row (4-byte aligned)
    long long base;                        // bit position of words[0], a multiple of 32
};
// the complete words below `pos` go out (with the partial one when `final_`), the partial one becomes word 0; wave-wide
__device__ __forceinline__ void wsink_flush(WaveSink& s, long long pos, bool final_) {
    FRAD_LDS_BARRIER();
    const int lane = threadIdx.x & 63;
    const int full = (int)((pos - s.base) >> 5), part = (int)((pos - s.base) & 31);
    const int n = final_ && part ? full + 1 : full;
    uint32_t* dst = reinterpret_cast<uint32_t*>(s.out + (s.base >> 3));
    const int carry = part && !final_ ? s.words[full] : 0;
    FRAD_LDS_BARRIER();                                       // (every lane has read the carry word before it is cleared)
    for (int i = lane; i <= full; i += 64) {
        const uint32_t v = (uint32_t)s.words[i];
        s.words[i] = (i == 0) ? carry : 0;
        if (i < n) dst[i] = bswap32(v);
    }
    s.base += (long long)full * 32;
    FRAD_LDS_BARRIER();
}
// one lane's stretch of the stream: bits are appended to a 64-bit accumulator whose top `nacc` (< 32) bits are taken
struct LaneBits {
    u64 acc; int nacc, w;
    __device__ __forceinline__ void put(int* words, uint32_t bits, int nb) {          // nb <= 32 bits, `bits` < 2^nb
        if (nb <= 0) return;
        acc |= (u64)bits << (64 - nacc - nb);
        nacc += nb;
        if (nacc >= 32) { atomicOr(&words[w], (int)(uint32_t)(acc >> 32)); acc <<= 32; nacc -= 32; ++w; }
    }
    __device__ __forceinline__ void done(int* words) { if (nacc > 0 && (uint32_t)(acc >> 32)) atomicOr(&words[w], (int)(uint32_t)(acc >> 32)); }
};

// one stream (n values at `data`) appended at bit position `pos` (byte aligned): k byte + codes + zero padding.  k needs the
// maximum of the whole stream first: up to 4096 values (a stereo frame of 2048) stay in registers between the two looks --
// 64 per lane -- so the stream is read from memory once (the second read missed L2: 554 MB of traffic per 15 000 frames for
// 296 MB of algorithmic bytes); longer streams are read twice.
constexpr int GTR = 4;                     // tiles a lane keeps in registers
__device__ __forceinline__ void load_tile(const int32_t* __restrict__ data, long long n, long long j0, bool vec, int32_t (&v)[GV]) {
    if (vec && j0 + GV <= n) {
#pragma unroll
        for (int i = 0; i < GV; i += 4) {
            const v4u q4 = *FRAD_GCPTR(v4u, data + j0 + i);
            v[i] = (int32_t)q4[0]; v[i + 1] = (int32_t)q4[1]; v[i + 2] = (int32_t)q4[2]; v[i + 3] = (int32_t)q4[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < GV; ++i) v[i] = j0 + i < n ? data[j0 + i] : 0;
    }
}
__device__ __forceinline__ long long encode_stream_wave(WaveSink& s, long long pos, const int32_t* __restrict__ data, long long n) {
    const int lane = threadIdx.x & 63;
    const bool vec = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
    const bool inreg = n <= (long long)GTR * WTILE;           // (uniform)
    int32_t vr[GTR][GV];
    u64 dmax = 0;
    if (inreg) {
#pragma unroll
        for (int t = 0; t < GTR; ++t) {
            if ((long long)t * WTILE < n) load_tile(data, n, (long long)t * WTILE + lane * GV, vec, vr[t]);
            else {
#pragma unroll
                for (int i = 0; i < GV; ++i) vr[t][i] = 0;
            }
#pragma unroll
            for (int i = 0; i < GV; ++i) { const long long v = vr[t][i]; const u64 a = (u64)(v < 0 ? -v : v); dmax = a > dmax ? a : dmax; }
        }
    } else {
        for (long long i = lane; i < n; i += 64) { const long long v = data[i]; const u64 a = (u64)(v < 0 ? -v : v); dmax = a > dmax ? a : dmax; }
    }
    dmax = wave_max_u64(dmax);
    const int k = rice_k(dmax);
    if (lane == 0) atomicOr(&s.words[(int)((pos - s.base) >> 5)], (int)((uint32_t)k << (24 - (int)((pos - s.base) & 31))));   // pos is byte aligned
    pos += 8;
    auto tile = [&](long long t0, const int32_t (&v)[GV]) {
        const long long j0 = t0 + (long long)lane * GV;
        u64 code[GV]; int mine = 0;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            code[i] = zigzag(v[i]) + (1ull << k);
            // m = L - (k + 1) zeros, then the L-bit number, L = k + 1 or k + 2: k + 1 or k + 3 bits
            mine += j0 + i < n ? ((code[i] >> (k + 1)) ? k + 3 : k + 1) : 0;
        }
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = (int)__shfl((unsigned long long)(unsigned)incl, lane >= off ? lane - off : lane, 64);
            incl += lane >= off ? o : 0;
        }
        const int total = (int)__shfl((unsigned long long)(unsigned)incl, 63, 64);
        const int rel = (int)(pos - s.base) + incl - mine;
        LaneBits lb{0, rel & 31, rel >> 5};
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            if (j0 + i < n) {
                const int len = (code[i] >> (k + 1)) ? k + 3 : k + 1;
                if (len > 32) { lb.put(s.words, (uint32_t)(code[i] >> 32), len - 32); lb.put(s.words, (uint32_t)code[i], 32); }
                else lb.put(s.words, (uint32_t)code[i], len);
            }
        }
        lb.done(s.words);
        pos += total;
        wsink_flush(s, pos, false);
    };
    if (inreg) {
#pragma unroll
        for (int t = 0; t < GTR; ++t) if ((long long)t * WTILE < n) tile((long long)t * WTILE, vr[t]);
    } else {
        for (long long t0 = 0; t0 < n; t0 += WTILE) {
            int32_t v[GV];
            load_tile(data, n, t0 + (long long)lane * GV, vec, v);
            tile(t0, v);
        }
    }
    return (pos + 7) & ~7LL;                                   // bitstr2bytes pads with zeros
}

__global__ void __launch_bounds__(64) k_gol_encode(const int32_t* __restrict__ q, const int32_t* __restrict__ tq, long long nq, long long ntq,
                                                   unsigned char* __restrict__ body, long long stride, long long* __restrict__ nbytes) {
    FRAD_DYN_SMEM(smem);                                      // GOL_LDS bytes: the word buffer
    int* words = reinterpret_cast<int*>(smem);
    const long long f = blockIdx.x;
    WaveSink s{words, body + f * stride, 0};
    for (int i = threadIdx.x; i < GWORDS; i += 64) words[i] = 0;
    FRAD_LDS_BARRIER();
    // '>I' len(thres_gol) is only known once that stream is coded: it goes first, so code the thresholds from bit 32
    // on and patch the length in afterwards (the first word has left LDS by then)
    const long long end_t = encode_stream_wave(s, 32, tq + f * ntq, ntq);
    const long long end_q = encode_stream_wave(s, end_t, q + f * nq, nq);
    wsink_flush(s, end_q, true);
    if (threadIdx.x == 0) {
        const uint32_t tlen = (uint32_t)((end_t - 32) >> 3);
        *reinterpret_cast<uint32_t*>(s.out) = bswap32(tlen);
        nbytes[f] = end_q >> 3;
    }
}

// offsets[0] = 0, offsets[i + 1] = offsets[i] + nbytes[i]; one block: every thread sums a contiguous run of rows, one
// block scan over the 256 run totals, then every thread writes its run's prefix sums
__global__ void __launch_bounds__(GT) k_rows_scan(const long long* __restrict__ nbytes, long long n, long long* __restrict__ offsets) {
    FRAD_DYN_SMEM(smem);
    long long* tmp = reinterpret_cast<long long*>(smem);
    const long long per = (n + GT - 1) / GT;
    const long long a = (long long)threadIdx.x * per, e = a + per < n ? a + per : n;
    long long mine = 0;
    long long i = a;
    for (; i + 8 <= e; i += 8) {                                  // eight loads in flight (a thread's run is one dependent chain otherwise)
        long long v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = nbytes[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) mine += v[j];
    }
    for (; i < e; ++i) mine += nbytes[i];
    long long total;
    long long run = block_scan(mine, tmp, &total) - mine;        // sum of all rows before this thread's run
    if (threadIdx.x == 0) offsets[0] = 0;
    i = a;
    for (; i + 8 <= e; i += 8) {
        long long v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = nbytes[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) { run += v[j]; offsets[i + j + 1] = run; }
    }
    for (; i < e; ++i) { run += nbytes[i]; offsets[i + 1] = run; }
}
__global__ void __launch_bounds__(GT) k_rows_gather(const unsigned char* __restrict__ rows, long long stride, const long long* __restrict__ offsets,
                                                    unsigned char* __restrict__ out) {
    const long long f = blockIdx.x, a = offsets[f], n = offsets[f + 1] - a;
    const unsigned char* src = rows + f * stride;
    unsigned char* dst = out + a;
    // rows are 4-byte aligned, the packed position is not: align the destination, then copy words through a funnel shift
    long long i = threadIdx.x;
    const int head = (int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
    if (i < head && i < n) dst[i] = src[i];
    const long long words = n > head ? (n - head) >> 2 : 0;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
    const int sh = head * 8;
    for (long long w = threadIdx.x; w < words; w += blockDim.x) {
        const uint32_t lo = s32[w], hi = sh ? s32[w + 1] : 0;  // s32[w + 1] stays inside the row: stride covers the bound + 4
        d32[w] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    }
    for (long long b = head + words * 4 + threadIdx.x; b < n; b += blockDim.x) dst[b] = src[b];
}

// ---- decode --------------------------------------------------------------------------------------------------------
// One lane per frame, one wave per 64 frames.  A prefix code's boundaries are a sequential dependency, so the
// parallelism is across frames; what must not happen is 64 lanes each waiting for their own global load (a wave stalls
// as a whole).  So the wave moves the bitstreams in lock-step rounds: it copies the next 256 bytes of every lane's
// stream into that lane's LDS row (coalesced: one row per load instruction), then every lane parses its row to the end
// -- a resumable state machine, since a code may straddle two rounds -- and parks until the next round.
constexpr int DCW = 64;                    // 32-bit words of a stream per round and lane
constexpr int DPITCH = DCW + 1;            // row pitch in words: odd, so equal offsets of different rows hit different banks
constexpr int DEC_LDS = 64 * DPITCH * 4;

__device__ __forceinline__ int32_t sat32(long long v) { return v > 2147483647LL ? 2147483647 : v < -2147483648LL ? (int32_t)(-2147483647 - 1) : (int32_t)v; }
__device__ __forceinline__ bool wave_any(bool v) { return wave_allreduce_u64(v ? 1ull : 0ull, [](u64 a, u64 b) { return a | b; }) != 0; }

// the calling lane's stream: bytes [p, p + len) = k byte + code bits; out[0 .. cap) receives the values, zero-filled
// (inlined, and the LDS taken from the kernel's own symbol: a pointer that went through a real call is generic, FLAT
// accesses then wait on the global stores in flight as well)
__device__ __forceinline__ void decode_stream_wave(const unsigned char* p, long long len, int32_t* __restrict__ out, long long cap) {
    FRAD_DYN_SMEM(smem_);
    uint32_t* rows = reinterpret_cast<uint32_t*>(smem_);
    const int lane = threadIdx.x & 63;
    uint32_t* row = rows + lane * DPITCH;
    // ---- this lane's stream, seen as aligned 32-bit words
    const int k = len >= 1 ? (int)p[0] : 0;
    const unsigned char* bits = p + 1;
    const long long nbytes = len >= 1 ? len - 1 : 0;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(bits);
    const int lead = (int)(addr & 3);
    const unsigned char* a0 = bits - lead;
    const long long nwords = nbytes > 0 ? (lead + nbytes + 3) >> 2 : 0;
    long long bits_left = nbytes * 8;                        // stream bits not yet moved into the window
    u64 win = 0; int have = 0;                                // the top `have` bits of `win` are the next unread bits
    // ---- parser state (survives the rounds)
    int phase = 0; long long m = 0, want = 0; u64 val = 0; bool big = false;
    bool active = nbytes > 0 && cap > 0;
    long long n_out = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(out) & 15) == 0;    // four values per 16-byte store
    int32_t o0 = 0, o1 = 0, o2 = 0;
    for (long long round = 0; wave_any(active); ++round) {
        // ---- fill: row r <- words [round * DCW, +DCW) of lane r's stream, big-endian to MSB-first; 16 rows' loads in
        // flight at a time (one load instruction per row: 64 lanes x 4 bytes, coalesced)
        for (int r0 = 0; r0 < 64; r0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const u64 base = __shfl((u64)reinterpret_cast<uintptr_t>(a0), r0 + i, 64);
                const long long nw = (long long)__shfl((u64)(active ? nwords : 0), r0 + i, 64);
                const long long w = round * DCW + lane;
                v[i] = 0;
                if (w < nw) v[i] = *FRAD_GCPTR(uint32_t, reinterpret_cast<const unsigned char*>((uintptr_t)base) + 4 * w);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) rows[(r0 + i) * DPITCH + lane] = bswap32(v[i]);
        }
        FRAD_LDS_BARRIER();
        // ---- parse this lane's row
        int rw = 0;                                               // next word of the row
        const long long row_words = nwords - round * DCW < DCW ? nwords - round * DCW : DCW;
        auto refill = [&]() {
            while (have <= 32 && rw < row_words) {
                uint32_t w = row[rw++];
                int valid = 32;
                if (round == 0 && rw == 1 && lead) { w <<= 8 * lead; valid -= 8 * lead; }      // bytes before the stream
                if (valid > bits_left) { valid = (int)bits_left; w = valid ? w & ~(0xffffffffu >> valid) : 0u; }   // bytes after it
                win |= (u64)w << (32 - have);
                have += valid; bits_left -= valid;
            }
        };
        auto emit = [&](int32_t v32) {
            const int slot = (int)(n_out & 3);
            if (!vec) out[n_out] = v32;
            else if (slot == 0) o0 = v32; else if (slot == 1) o1 = v32; else if (slot == 2) o2 = v32;
            else { const v4u q4 = {(uint32_t)o0, (uint32_t)o1, (uint32_t)o2, (uint32_t)v32}; *FRAD_GPTR(v4u, out + n_out - 3) = q4; }
            ++n_out;
            if (n_out >= cap) active = false;
        };
        while (active) {
            refill();
            if (phase == 0 && m == 0 && k < 31 && (win >> 32) != 0) {
                // the common case in straight-line 32-bit arithmetic: the whole code (z zeros, then z + k + 1 bits) lies in
                // the top 32 bits of the window
                const uint32_t top = (uint32_t)(win >> 32);
                const int z = __builtin_clz(top), nb = z + k + 1, total = z + nb;
                if (total <= 32 && total <= have) {
                    const int32_t n = (int32_t)((top << z) >> (32 - nb)) - (int32_t)(1u << k);       // >= 0: the leading bit is 2^(z+k)
                    emit((n & 1) ? (n + 1) >> 1 : -(n >> 1));
                    win <<= total; have -= total;
                    continue;
                }
            }
            if (phase == 0) {                                     // m = index of the first '1' (p1tools.py:68)
                if (have == 0) { if (bits_left == 0) active = false; break; }      // none left: stop -- or next round
                if (win == 0) { m += have; have = 0; continue; }
                const int z = __builtin_clzll(win);               // < have: only the top `have` bits can be set
                m += z; win = z >= 64 ? 0 : win << z; have -= z;
                // the codeword is data[:2m + k + 1]: the m zeros just skipped and m + k + 1 more bits -- fewer at the end
                // of the buffer, where Python's slice is simply shorter
                want = m + (long long)k + 1;
                const long long left = (long long)have + bits_left;
                if (want > left) want = left;
                val = 0; big = false; phase = 1;
            }
            bool starved = false;
            while (want > 0) {
                refill();
                if (have == 0) { starved = true; break; }         // the rest of this code arrives with the next round
                int take = want > 32 ? 32 : (int)want;
                if (take > have) take = have;
                if (val >> (64 - take)) big = true;
                val = (val << take) | (win >> (64 - take));
                win = take >= 64 ? 0 : win << take; have -= take; want -= take;
            }
            if (starved) break;
            // n = int(codeword, 2) - 2^k; value = (n + 1) >> 1 if n is odd else -(n >> 1)   (Python integers)
            long long v;
            if (big || k >= 62 || val >= (1ull << 62)) {
                // beyond anything an int32 quantiser emits (a corrupt stream): keep the sign rule, saturate the size
                const bool n_neg = !big && (k >= 64 || (k >= 62 && val < (1ull << k)));
                const bool odd = ((val & 1ull) != 0) != (k == 0);
                v = (odd != n_neg) ? 0x7fffffffffffLL : -0x7fffffffffffLL;
            } else {
                const long long n = (long long)val - (1LL << k);
                v = (n & 1) ? (n + 1) >> 1 : -(n >> 1);
            }
            emit(sat32(v));
            phase = 0; m = 0;
        }
        FRAD_LDS_BARRIER();                                       // rows are rewritten next round
    }
    if (vec && (n_out & 3)) {                                 // values parked in registers: write them singly
        const long long base = n_out & ~3LL; const int cnt = (int)(n_out & 3);
        out[base] = o0; if (cnt > 1) out[base + 1] = o1; if (cnt > 2) out[base + 2] = o2;
    }
    if (vec) {
        for (; n_out < cap && (n_out & 3); ++n_out) out[n_out] = 0;
        const v4u z4 = {0, 0, 0, 0};
        for (; n_out + 4 <= cap; n_out += 4) *FRAD_GPTR(v4u, out + n_out) = z4;
    }
    for (; n_out < cap; ++n_out) out[n_out] = 0;
}

__global__ void __launch_bounds__(64) k_gol_decode(const unsigned char* __restrict__ bodies, const long long* __restrict__ offsets, long long n_frames,
                                                   long long nq, long long ntq, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                                   int32_t* __restrict__ status, const int32_t* __restrict__ todo) {
    const long long f = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = f < n_frames;                           // spare lanes of the last wave keep the rounds uniform
    const unsigned char* b = live ? bodies + offsets[f] : bodies;
    long long len = live ? offsets[f + 1] - offsets[f] : 0;
    long long tlen = 0;
    const int mine = live ? (todo ? todo[f] : 3) : 0;         // streams the wave-per-frame kernel left to this one (bit 0: tq, bit 1: q)
    if (live && status && !todo) status[f] = len < 4 ? 1 : 0;          // no length word: nothing decodable (the host treats it as broken)
    if (len >= 4) {
        tlen = ((long long)b[0] << 24) | ((long long)b[1] << 16) | ((long long)b[2] << 8) | (long long)b[3];
        if (tlen > len - 4) tlen = len - 4;                   // frad[:thresbytes] past the end: Python slicing clamps
    } else len = 4;                                           // both streams empty -> zero fill
    for (int which = 0; which < 2; ++which) {                 // thresholds, then coefficients (one copy of the parser)
        const unsigned char* sp = which ? b + 4 + tlen : b + 4;
        const long long sl = which ? len - 4 - tlen : tlen;
        int32_t* dst = which ? q + (live ? f * nq : 0) : tq + (live ? f * ntq : 0);
        decode_stream_wave(sp, sl, dst, (mine >> which) & 1 ? (which ? nq : ntq) : 0);
    }
}

// ---- decode, one WAVE per frame (the fast path) ----------------------------------------------------------------------
// The stream sits in LDS as big-endian words; lane i owns the bits [i CHB, (i+1) CHB), CHB an odd number of words.  A prefix
// code's boundaries are a sequential dependency.  Two ways around it, two instantiations of one kernel:
//   MAP = false (every frame): SPECULATE -- all lanes at once walk the code lengths of their chunk from a guessed entry (its
//     first bit) and note where the walk leaves the chunk; every lane then takes its left neighbour's exit as its entry and
//     walks again if that differs from what it assumed, until no lane changes (lane 0's entry is known, so lane i is right
//     after at most i rounds).  The walks are lenient (see `walk`), which is what makes tonal frames settle in a round or
//     two; whatever realigned is walked again exactly.  ~20 instructions per code and walk.
//   MAP = true (the streams the walks did not settle in 16 rounds, `todo`): the entry map of EVERY bit position of the chunk
//     by a backward dynamic programme (21 instructions x 392 positions per lane whatever the data), then 64 chained look-ups.
// Phase 2: output index of each chunk's first code.  Phase 3, all lanes at once: each decodes exactly its own codes from a
// 64-bit window over the LDS words.  Codes longer than 32 bits, k > 30 and streams beyond the LDS budget are left to the
// lane-per-frame kernel (`todo`).
constexpr int GW_WORDS = 6144;             // most stream words a wave holds in LDS (24 KiB); the launch sizes it for 16 bits per value
constexpr int GW_E = 32;                   // longest code / entry range handled here
constexpr int GW_RING = 64;                // (entry maps) ring entries per lane (>= GW_E + the four positions of a batch)
constexpr int GW_PITCH = 66;               // ring pitch in 16-bit entries (33 words: odd, lanes hit different banks)
constexpr int gw_lds(int wmax, int omax, bool map = false) { return (wmax + 2 + omax) * 4 + (map ? 64 * GW_PITCH * 2 : 0); }
constexpr uint32_t GW_END = 127, GW_LONG = 126;          // a walk's exit: offset 0 .. 31 into the next chunk, or one of these

// one stream of the calling WAVE's frame; false = leave it to the slow kernel (nothing written)
template <bool MAP>
__device__ __forceinline__ bool decode_stream_2phase(const unsigned char* p, long long len, int32_t* __restrict__ out, long long cap, int wmax, int omax) {
    FRAD_DYN_SMEM(smem_);
    uint32_t* words = reinterpret_cast<uint32_t*>(smem_);
    uint32_t* obuf = words + wmax + 2;                                // the decoded values on their way out (omax words)
    const bool staged = cap + (cap >> 5) + 1 <= (long long)omax;      // (uniform)
    const int lane = threadIdx.x & 63;
    const long long nbytes = len >= 1 ? len - 1 : 0;
    long long total = 0;
    if (nbytes > 0 && cap > 0) {
        const int k = (int)p[0];
        const unsigned char* bits = p + 1;
        const int lead = (int)(reinterpret_cast<uintptr_t>(bits) & 3);
        const unsigned char* a0 = bits - lead;
        const long long nwords_ll = (lead + nbytes + 3) >> 2;
        if (nwords_ll > wmax || k > 30) return false;
        const int nwords = (int)nwords_ll;
        const int T = lead * 8 + (int)nbytes * 8;                 // end of the stream in word space; it starts at bit lead * 8
        FRAD_LDS_BARRIER();                                       // the previous stream's phases are done with the LDS
        int last1 = -1;                                           // position of the stream's last '1'
        for (int w = lane; w < nwords + 2; w += 64) {
            uint32_t v = 0;
            if (w < nwords) {
                v = bswap32(*FRAD_GCPTR(uint32_t, a0 + 4 * w));
                if (w == 0 && lead) v &= 0xffffffffu >> (8 * lead);          // bytes before the stream
                const int hi = (w + 1) * 32;
                if (hi > T) v = (T - w * 32) > 0 ? v & ~(0xffffffffu >> (T - w * 32)) : 0u;      // bytes after it
            }
            words[w] = v;
            if (v) { const int p1 = w * 32 + 31 - __builtin_ctz(v); last1 = p1 > last1 ? p1 : last1; }
        }
        for (int off = 1; off < 64; off <<= 1) { const int o = (int)__shfl_xor((unsigned long long)(unsigned)last1, off, 64); last1 = o > last1 ? o : last1; }
        FRAD_LDS_BARRIER();
        int chb = ((T + 63) / 64 + 31) & ~31;                     // whole words per chunk,
        if (chb < 32) chb = 32;
        if (!((chb >> 5) & 1)) chb += 32;                         // an odd number of them: the lanes' window reads hit different banks
        const int nch = (T + chb - 1) / chb;
        const int cs = lane * chb;
        int my_base = 0, my_cnt = 0, my_entry = 0;
        if constexpr (MAP) {
            // The exhaustive form (rounds 2 / 3 of this decoder; now behind the walks, for the streams they do not settle): the entry map of
            // EVERY bit position of the chunk by a backward dynamic programme, entry(p) = entry(p + 2 z(p) + k + 1) + one code, in a
            // ring of the last 64 positions (16-bit entries: exit offset or END / LONG in the low 7 bits, codes above); a code is at
            // least k + 1 bits long, so for k >= 3 four positions are taken per LDS round trip.  Then 64 dependent look-ups chain the maps.
            if (chb / (k + 1) > 511) return false;                // the 9-bit code count of an entry
            unsigned short* rings = reinterpret_cast<unsigned short*>(obuf + omax);
            unsigned short* ring = rings + lane * GW_PITCH;
            // ---- phase 1: the entry map of this lane's chunk ---------------------------------------------------
            const int ce = cs + chb;
            if (lane < nch) {
                // Everything in chunk-relative positions r = pp - cs.  zcap = first zero run too long for a 32-bit code ("zcap or more").
                const int zcap = (GW_E - k - 1) / 2 + 1;
                const int rl1 = last1 - cs, rT = T - cs;                      // last '1' and stream end, relative to this chunk
                int z;
                {
                    const int wn = ce >> 5;                                   // (the last chunks may reach beyond the staged words: zeros)
                    const uint32_t w0 = wn < nwords + 2 ? words[wn] : 0u, w1 = wn + 1 < nwords + 2 ? words[wn + 1] : 0u;
                    z = w0 ? __builtin_clz(w0) : (w1 ? 32 + __builtin_clz(w1) : 64);
                    if (z > zcap) z = zcap;
                }
                // entry of position r given the zero run z there: either final (`e`, ri < 0) or one more code behind ring slot `ri`.
                // Branch-free (selects): the lanes of a wave sit in different cases at every position.
                auto classify = [&](int r, int z_, uint32_t& e, int& ri) {
                    const int rnx = r + 2 * z_ + k + 1;                               // where the next code starts
                    const bool end_ = r > rl1, lng = z_ >= zcap, lastc = rnx >= rT, outc = rnx >= chb;
                    uint32_t v = (uint32_t)(rnx - chb) | (1u << 7);                   // the code ends in the next chunk (exit offset < 32)
                    v = lastc ? ((1u << 7) | GW_END) : v;                             // the stream's last code (maybe cut short)
                    v = lng ? GW_LONG : v;
                    v = end_ ? GW_END : v;                                            // zeros to the end: no code starts here
                    e = v;
                    const int slot = rnx & (GW_RING - 1);                             // (cs is a multiple of 32 and the ring index is taken mod 64 of r)
                    ri = (end_ | lng | lastc | outc) ? (slot | (int)0x80000000) : slot;     // sign bit: the ring is not consulted
                };
                // one more code in front of entry t: the count sits above bit 7, so a LONG / END code in the low 7 bits rides along
                auto finish = [&](uint32_t e, int ri, uint32_t t) -> uint32_t { return ri < 0 ? e : t + (1u << 7); };
                for (int w = (ce >> 5) - 1; w >= (cs >> 5); --w) {
                    const uint32_t cur = w < nwords + 2 ? words[w] : 0u;
                    const int r0 = 32 * w + 31 - cs;                                  // relative position of the word's last bit
                    if (k >= 3) {
    #pragma unroll 2
                        for (int i = 0; i < 32; i += 4) {                 // positions r0 - i ... - 3: mutually independent (k + 1 >= 4)
                            uint32_t e[4]; int ri[4]; uint32_t t[4];
    #pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool one = (cur >> (i + j)) & 1u;
                                z = one ? 0 : (z < zcap ? z + 1 : zcap);
                                classify(r0 - i - j, z, e[j], ri[j]);
                            }
    #pragma unroll
                            for (int j = 0; j < 4; ++j) t[j] = ring[ri[j] & (GW_RING - 1)];
    #pragma unroll
                            for (int j = 0; j < 4; ++j) ring[(r0 - i - j) & (GW_RING - 1)] = (unsigned short)finish(e[j], ri[j], t[j]);
                        }
                    } else {
                        for (int i = 0; i < 32; ++i) {
                            const bool one = (cur >> i) & 1u;
                            z = one ? 0 : (z < zcap ? z + 1 : zcap);
                            uint32_t e; int ri;
                            classify(r0 - i, z, e, ri);
                            const uint32_t t = ring[ri & (GW_RING - 1)];
                            ring[(r0 - i) & (GW_RING - 1)] = (unsigned short)finish(e, ri, t);
                        }
                    }
                }
            }
            FRAD_LDS_BARRIER();
            // ---- phase 2: chain the maps (every lane walks the same chain and keeps its own link) -----------------
            {
                int e = lead * 8, base = 0; bool ended = false, bad = false;
                for (int i = 0; i < nch; ++i) {
                    if (!ended) {
                        const uint32_t t = rings[i * GW_PITCH + e];
                        const uint32_t x = t & 127u;
                        if (x == GW_LONG) { bad = true; break; }
                        if (i == lane) { my_base = base; my_entry = e; my_cnt = (int)(t >> 7); }
                        base += (int)(t >> 7);
                        if (x == GW_END) ended = true; else e = (int)x;
                    }
                }
                if (bad) return false;                                                // (uniform: every lane read the same entries)
                total = base < cap ? base : cap;
            }
        } else {
            // ---- phase 1: where does this lane's chunk begin?  (speculate, then correct) ---------------------------
            // Chunk-relative positions r = pp - cs.  zcap = first zero run too long for a 32-bit code ("zcap or more").
            const int zcap = (GW_E - k - 1) / 2 + 1;
            const int rl1 = last1 - cs, rT = T - cs;                  // last '1' and stream end, relative to this chunk
            const bool mine = lane < nch;
            // walk the codes from chunk-relative position e to the first one that starts beyond the chunk: x = its offset there
            // (< GW_E), GW_END (the stream ended: the last code may be cut short, or only zeros were left) or GW_LONG (a code this
            // path does not take); cnt = codes that start in the chunk.  Selects, not branches: the lanes differ at every step.
            // `lenient`: with the k of the reference coder (taken from the stream's maximum, p1tools.py:47-48) a code has at most ONE
            // leading zero, so two or more zeros say "this is not a code boundary" and a lenient walk moves on to the next '1' (and
            // notes that it did).  Without that, a run of equal values -- the zeros of a tonal frame's upper bands: '1 0^k' repeated
            // -- keeps a wrong start out of step for the whole chunk, and the corrections below advance one lane per round (40-50
            // rounds measured on harmonic test signals; one or two with the realignment).  A walk that never realigned IS the
            // exact walk; only those count in the end.
            auto walk = [&](int e, bool act, int& x, int& cnt, bool lenient, bool& realigned) {
                bool go = act && e < GW_E;
                if (act) { x = e < GW_E ? (int)GW_END : e; cnt = 0; realigned = false; }  // an END / LONG entry is passed on
                int r = e & (GW_E - 1);
                const int pos = cs + r;
                int wi = pos >> 5;
                u64 win = 0; int have = 0;
                if (go) { win = (((u64)words[wi] << 32) | (u64)words[wi + 1]) << (pos & 31); have = 64 - (pos & 31); wi += 2; }
                while (go) {
                    if (have < 32) { win |= (u64)words[wi < nwords + 2 ? wi : nwords + 1] << (32 - have); have += 32; ++wi; }
                    const int z = __builtin_clz((uint32_t)(win >> 32) | 1u);
                    const bool end_ = r > rl1;
                    const bool re = lenient && z >= 2 && !end_;                       // realign: skip the zeros, count nothing
                    const int used = re ? z : 2 * z + k + 1, rn = r + used;
                    const bool lng = !re && z >= zcap, lastc = rn >= rT, outc = rn >= chb;
                    int v = x;
                    v = outc ? rn - chb : v;
                    v = lastc ? (int)GW_END : v;
                    v = lng ? (int)GW_LONG : v;
                    v = end_ ? (int)GW_END : v;
                    x = v;
                    cnt += (end_ | lng | re) ? 0 : 1;
                    realigned |= re;
                    go = !(end_ | lng | lastc | outc);
                    win <<= (used & 63); have -= used; r = rn;            // (used <= 32 while `go` stays set)
                }
            };
            my_entry = lane == 0 ? lead * 8 : 0; my_cnt = 0;
            int my_x = (int)GW_END;
            bool my_re = false;
            // Round 0: every chunk from its first bit -- a guess; prefix codes fall into step within a few codes, so most exits are right
            // already.  Then: lane i takes lane i - 1's exit as its entry and walks again if that differs from what it assumed.  Lane
            // 0's entry is known (its walks are exact), so lane i is final once lane i - 1 is.
            walk(my_entry, mine, my_x, my_cnt, lane > 0, my_re);
            for (int round = 0;; ++round) {
                const int ex = (int)__shfl((unsigned long long)(unsigned)my_x, lane ? lane - 1 : 0, 64);
                const bool redo = mine && lane > 0 && ex != my_entry;
                if (!wave_any(redo)) break;
                if (round >= 16) return false;                            // (uniform) a stream that keeps its chunks out of step: slow kernel
                if (redo) my_entry = ex;
                walk(my_entry, redo, my_x, my_cnt, true, my_re);
            }
            // Every lane now starts where its left neighbour ended.  A lane whose last walk realigned has not walked the code as it is
            // written (a stream with longer zero prefixes than the reference coder makes, or a damaged one): those walk again, exactly,
            // and the corrections repeat with exact walks.
            if (wave_any(mine && my_re)) {
                bool dummy = false;
                walk(my_entry, mine && my_re, my_x, my_cnt, false, dummy);
                for (int round = 0;; ++round) {
                    const int ex = (int)__shfl((unsigned long long)(unsigned)my_x, lane ? lane - 1 : 0, 64);
                    const bool redo = mine && lane > 0 && ex != my_entry;
                    if (!wave_any(redo)) break;
                    if (round >= 16) return false;
                    if (redo) my_entry = ex;
                    walk(my_entry, redo, my_x, my_cnt, false, dummy);
                }
            }
            // ---- phase 2: output index of every chunk's first code ------------------------------------------------------
            if (wave_any(mine && my_x == (int)GW_LONG)) return false;     // (uniform) a code beyond 32 bits on the path
            if (!mine) my_cnt = 0;
            my_base = my_cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = (int)__shfl((unsigned long long)(unsigned)my_base, lane >= off ? lane - off : lane, 64);
                my_base += lane >= off ? o : 0;
            }
            {
                const int sum = (int)__shfl((unsigned long long)(unsigned)my_base, 63, 64);
                total = sum < cap ? sum : cap;
            }
            my_base -= my_cnt;
        }
        // ---- phase 3: every lane decodes the codes that start in its chunk ------------------------------------
        // The values go to LDS first when the frame fits (value i at word i + i / 32: the lanes' output ranges begin about 64
        // values apart), and leave as whole 256-byte / 1 KiB rows -- 64 lanes storing 4 bytes each at 64 places were the
        // slowest part of this phase.
        if (lane < nch && my_cnt > 0) {                               // exactly the codes phase 1 counted for this chunk: each has its
            int pos = cs + my_entry;                                  // '1' before T and is at most GW_E bits long
            int idx = my_base;
            int wi = pos >> 5;
            // 64-bit window, left-aligned at `pos`; `have` valid bits (the words beyond the stream are zero)
            u64 win = (((u64)words[wi] << 32) | (u64)words[wi + 1]) << (pos & 31);
            int have = 64 - (pos & 31);
            wi += 2;
            const int stop = (long long)my_base + my_cnt < cap ? my_base + my_cnt : (int)cap;
            const bool vec = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
            int32_t o0 = 0, o1 = 0, o2 = 0;
            for (; idx < stop; ++idx) {
                if (have < 32) { win |= (u64)words[wi < nwords + 2 ? wi : nwords + 1] << (32 - have); have += 32; ++wi; }
                const uint32_t top = (uint32_t)(win >> 32);
                const int z = __builtin_clz(top | 1u);                          // <= 15 here (the code fits 32 bits)
                int want = z + k + 1;                                           // bits from the '1' on: <= 31 (k <= 30)
                const int left = T - (pos + z);
                if (want > left) want = left;                                   // cut short by the end of the buffer
                const int used = z + want;
                const uint32_t val = want > 0 ? (top << z) >> (32 - want) : 0u;
                win <<= used; have -= used; pos += used;
                const int32_t n = (int32_t)val - (int32_t)(1u << k), hf = n >> 1;       // val < 2^31, 2^k <= 2^30
                const int32_t v = (n & 1) ? hf + 1 : -hf;                       // (n + 1) >> 1 if n is odd else -(n >> 1)
                if (staged) obuf[idx + (idx >> 5)] = (uint32_t)v;
                else {
                    // straight to memory, four values per store where a whole aligned group lies in this lane's range (a
                    // quarter of the write requests: 64 lanes x 4 bytes at 64 places is what bounds this phase)
                    const int g0 = idx & ~3, slot = idx & 3;
                    if (!vec || g0 < my_base || g0 + 3 >= stop) out[idx] = v;
                    else if (slot == 0) o0 = v; else if (slot == 1) o1 = v; else if (slot == 2) o2 = v;
                    else { const v4u q4 = {(uint32_t)o0, (uint32_t)o1, (uint32_t)o2, (uint32_t)v}; *FRAD_GPTR(v4u, out + g0) = q4; }
                }
            }
        }
    }
    if (staged) {
        FRAD_LDS_BARRIER();
        const int tot = (int)total, n = (int)cap;
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            for (int j = lane * 4; j < n; j += 256) {
                const uint32_t* o = obuf + j + (j >> 5);                        // four values of one group of 32: adjacent words
                if (j + 4 <= n) {
                    const v4u q4 = {j < tot ? o[0] : 0u, j + 1 < tot ? o[1] : 0u, j + 2 < tot ? o[2] : 0u, j + 3 < tot ? o[3] : 0u};
                    *FRAD_GPTR(v4u, out + j) = q4;
                } else for (int i = 0; j + i < n; ++i) out[j + i] = j + i < tot ? (int32_t)o[i] : 0;
            }
        } else for (int j = lane; j < n; j += 64) out[j] = j < tot ? (int32_t)obuf[j + (j >> 5)] : 0;
        return true;
    }
    for (long long j = total + lane; j < cap; j += 64) out[j] = 0;
    return true;
}

// MAP = false: every frame, the walks.  MAP = true: the frames (streams) the walks left in `todo`, with the entry maps; what
// neither takes stays in `todo` for the lane-per-frame kernel.
template <bool MAP>
__global__ void __launch_bounds__(64) k_gol_decode_wave(const unsigned char* __restrict__ bodies, const long long* __restrict__ offsets,
                                                        long long nq, long long ntq, int32_t* __restrict__ q, int32_t* __restrict__ tq,
                                                        int32_t* __restrict__ status, int32_t* __restrict__ todo, int wmax, int omax) {
    const long long f = blockIdx.x;
    int mine = 3;
    if constexpr (MAP) { mine = todo[f]; if (mine == 0) return; }                  // (uniform: one wave, one frame)
    const unsigned char* b = bodies + offsets[f];
    long long len = offsets[f + 1] - offsets[f];
    long long tlen = 0;
    if (!MAP && threadIdx.x == 0 && status) status[f] = len < 4 ? 1 : 0;
    if (len >= 4) {
        tlen = ((long long)b[0] << 24) | ((long long)b[1] << 16) | ((long long)b[2] << 8) | (long long)b[3];
        if (tlen > len - 4) tlen = len - 4;
    } else len = 4;
    const bool ok_t = (mine & 1) ? decode_stream_2phase<MAP>(b + 4, tlen, tq + f * ntq, ntq, wmax, omax) : true;
    const bool ok_q = (mine & 2) ? decode_stream_2phase<MAP>(b + 4 + tlen, len - 4 - tlen, q + f * nq, nq, wmax, omax) : true;
    if (threadIdx.x == 0) todo[f] = (ok_t ? 0 : 1) | (ok_q ? 0 : 2);
}

thread_local int g_gol_hip = 0;
#define GOLCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_gol_hip = (int)e_; return FRAD_E_HIP; } } while (0)

}  // namespace
int golomb_last_hip_error() { return g_gol_hip; }
}  // namespace frad

using namespace frad;

extern "C" {

size_t frad_p1_golomb_bound(int32_t N, int32_t C) {
    if (N < 1 || C < 1) return 0;
    const size_t nq = (size_t)N * (size_t)C, nt = 27 * (size_t)C;
    const size_t bytes = 4 + (1 + (nt * 35 + 7) / 8) + (1 + (nq * 35 + 7) / 8);
    return (bytes + 8 + 15) / 16 * 16;                        // + one word of slack for the gather's funnel read
}

int frad_p1_golomb_encode(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C,
                          void* bodies, int64_t body_stride, int64_t* body_bytes, void* stream) {
    if (n_frames < 0 || N < 1 || C < 1 || C > 256) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!q || !tq || !bodies || !body_bytes) return FRAD_E_INVALID;
    if (body_stride < (int64_t)frad_p1_golomb_bound(N, C) || (body_stride & 3) || (reinterpret_cast<uintptr_t>(bodies) & 3)) return FRAD_E_INVALID;
    if (n_frames > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_gol_encode, dim3((unsigned)n_frames), dim3(64), GOL_LDS, static_cast<hipStream_t>(stream), q, tq, (long long)N * C, 27LL * C,
                       static_cast<unsigned char*>(bodies), (long long)body_stride, reinterpret_cast<long long*>(body_bytes));
    GOLCHK(hipGetLastError());
    return FRAD_OK;
}

int frad_rows_compact(const void* rows, int64_t row_stride, const int64_t* row_bytes, int64_t n_rows, void* out, int64_t* offsets, void* stream) {
    if (n_rows < 0 || row_stride < 0) return FRAD_E_INVALID;
    if (!offsets) return FRAD_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_rows > 0 && (!rows || !row_bytes || (reinterpret_cast<uintptr_t>(rows) & 3) || (row_stride & 3) || row_stride < 4)) return FRAD_E_INVALID;
    if (n_rows > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(GT), GT * 8, s, reinterpret_cast<const long long*>(row_bytes), (long long)n_rows, reinterpret_cast<long long*>(offsets));
    if (n_rows > 0 && out)
        hipLaunchKernelGGL(k_rows_gather, dim3((unsigned)n_rows), dim3(GT), 0, s, static_cast<const unsigned char*>(rows), (long long)row_stride,
                           reinterpret_cast<const long long*>(offsets), static_cast<unsigned char*>(out));
    GOLCHK(hipGetLastError());
    return FRAD_OK;
}

// diagnostic (not part of the ABI): the wave-per-frame kernel alone; todo[f] bit 0 / 1 = the threshold / coefficient stream of
// frame f was left to the lane-per-frame kernel (tests pin which streams take the fast path)
int frad_debug_golomb_decode_wave(const void* bodies, const int64_t* offsets, int64_t n_frames, int32_t N, int32_t C,
                                  int32_t* q, int32_t* tq, int32_t* todo, int32_t with_maps, void* stream) {
    if (n_frames <= 0 || N < 1 || C < 1 || !bodies || !offsets || !q || !tq || !todo || n_frames > 0x7fffffffLL) return FRAD_E_INVALID;
    long long wmax = ((long long)N * C * 16) / 32 + 64;
    if (wmax > GW_WORDS) wmax = GW_WORDS;
    const long long nq = (long long)N * C, nst = nq <= 512 ? nq : 27LL * C;
    const int omax = (int)(nst + (nst >> 5) + 1);
    hipLaunchKernelGGL(k_gol_decode_wave<false>, dim3((unsigned)n_frames), dim3(64), gw_lds((int)wmax, omax), static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(bodies), reinterpret_cast<const long long*>(offsets), nq, 27LL * C, q, tq,
                       static_cast<int32_t*>(nullptr), todo, (int)wmax, omax);
    if (with_maps)
        hipLaunchKernelGGL(k_gol_decode_wave<true>, dim3((unsigned)n_frames), dim3(64), gw_lds((int)wmax, omax, true), static_cast<hipStream_t>(stream),
                           static_cast<const unsigned char*>(bodies), reinterpret_cast<const long long*>(offsets), nq, 27LL * C, q, tq,
                           static_cast<int32_t*>(nullptr), todo, (int)wmax, omax);
    GOLCHK(hipGetLastError());
    return FRAD_OK;
}

int frad_p1_golomb_decode(const void* bodies, const int64_t* offsets, int64_t n_frames, int32_t N, int32_t C,
                          int32_t* q, int32_t* tq, int32_t* status, void* stream) {
    if (n_frames < 0 || N < 1 || C < 1 || C > 256) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!bodies || !offsets || !q || !tq) return FRAD_E_INVALID;
    const long long blocks = ((long long)n_frames + 63) / 64;
    if (n_frames > 0x7fffffffLL) return FRAD_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const bool no_wave = [] { const char* e = std::getenv("FRAD_TUNE_GOLOMB_LANE"); return e && e[0] == '1'; }();
    int32_t* todo = nullptr;
    if (!no_wave) {
        // fast path: one wave per frame; what it cannot take (codes > 64 bits, k > 30, streams beyond its LDS) is marked in
        // `todo` and decoded by the lane-per-frame kernel behind it
        if (hipMallocAsync(reinterpret_cast<void**>(&todo), sizeof(int32_t) * (size_t)n_frames, s) != hipSuccess) return FRAD_E_NOMEM;
        // LDS budget: 16 bits per coefficient on average (a frame above that goes to the slow kernel): at N C = 4096 that is
        // 8.4 KiB of stream per wave
        long long wmax = ((long long)N * C * 16) / 32 + 64;
        if (wmax > GW_WORDS) wmax = GW_WORDS;
        // + a short stream's values on their way out (the thresholds; a frame of <= 512 coefficients).  Staging whole frames of
        // 4096 was measured: 25 KiB per wave leaves six waves on a CU -- 0.46 ms per 15 000 frames against 0.37 without
        const long long nq = (long long)N * C, nst = nq <= 512 ? nq : 27LL * C;
        const int omax = (int)(nst + (nst >> 5) + 1);
        hipLaunchKernelGGL(k_gol_decode_wave<false>, dim3((unsigned)n_frames), dim3(64), gw_lds((int)wmax, omax), s, static_cast<const unsigned char*>(bodies),
                           reinterpret_cast<const long long*>(offsets), (long long)N * C, 27LL * C, q, tq, status, todo, (int)wmax, omax);
        // behind it, for the streams the walks did not settle (a wave whose frame has none returns at once): the entry maps
        hipLaunchKernelGGL(k_gol_decode_wave<true>, dim3((unsigned)n_frames), dim3(64), gw_lds((int)wmax, omax, true), s, static_cast<const unsigned char*>(bodies),
                           reinterpret_cast<const long long*>(offsets), (long long)N * C, 27LL * C, q, tq, status, todo, (int)wmax, omax);
    }
    hipLaunchKernelGGL(k_gol_decode, dim3((unsigned)blocks), dim3(64), DEC_LDS, s, static_cast<const unsigned char*>(bodies),
                       reinterpret_cast<const long long*>(offsets), (long long)n_frames, (long long)N * C, 27LL * C, q, tq, status, todo);
    if (todo) (void)hipFreeAsync(todo, s);
    GOLCHK(hipGetLastError());
    return FRAD_OK;
}

}  // extern "C"
