// frad_persistent.hpp -- persistent, software-pipelined variants of the profile-0 FFT kernels for
// the headline geometries (N = 2048 float64, N = 4096 float32).
//
// Why: the one-shot kernels (frad_kernels.hpp) pay four dependent trips to L2/HBM per block --
// PCM in, twiddles of pass 2, of pass 3, of the DCT step -- with only two waves per SIMD to hide
// them (rocprof, round 1: SQ_WAIT_ANY = 50 % of wave cycles, wave lifetime 22 us for 7 us of
// issue).  Here one block per CU stays resident and loops over frame groups:
//   * the twiddle / DCT tables are copied to LDS once per block, so the transform itself never
//     touches global memory;
//   * the PCM (or payload) of group i+1 is loaded into registers while group i is transformed,
//     and the stores of group i drain while group i+1 is staged: HBM latency is off the critical
//     path and both directions of the memory pipe stay busy during the butterflies;
//   * LDS = tables (~30 KiB) + 8 teams x 16 KiB of swizzled FFT buffers = 158 of the 160 KiB.
#pragma once
#include "frad_kernels.hpp"
#include <type_traits>
#ifdef FRAD_HOST_EMULATION
#include <thread>
#endif

namespace frad {

// Plans of the persistent kernels: radices R1..R4 (0 = no such pass), lanes per channel-frame,
// swizzle shift (log2 R1).
struct PlanA10 { static constexpr int LOG2M = 10, TEAM = 64,  SH = 4, R1 = 16, R2 = 16, R3 = 4, R4 = 0; };  // 2 waves/SIMD
struct PlanB10 { static constexpr int LOG2M = 10, TEAM = 128, SH = 3, R1 = 8,  R2 = 8,  R3 = 8, R4 = 2; };  // 4 waves/SIMD
struct PlanI10 { static constexpr int LOG2M = 10, TEAM = 64,  SH = 100, R1 = 4, R2 = 16, R3 = 16, R4 = 0; };  // inverse, fused first pass
struct PlanA9  { static constexpr int LOG2M = 9,  TEAM = 64,  SH = 3, R1 = 8,  R2 = 8,  R3 = 8, R4 = 0; };   // N = 1024: two blocks per CU
struct PlanA11 { static constexpr int LOG2M = 11, TEAM = 64,  SH = 4, R1 = 16, R2 = 16, R3 = 8, R4 = 0; };

// resident waves per SIMD the unit kernels are compiled for: N = 1024 (8 KiB per channel-frame) lets two blocks share a CU
template <typename PL> struct UnitWaves { static constexpr int value = PL::LOG2M <= 9 ? 4 : 2; };

// LDS table blob of a plan (units: complex slots): [pass-2][pass-3][pass-4 tables][w_k][g_k]
template <typename PL> struct PersLayout {
    static constexpr int M = 1 << PL::LOG2M, TEAM = PL::TEAM;
    static constexpr int NS2 = PL::R1, NS3 = PL::R1 * PL::R2, NS4 = PL::R1 * PL::R2 * PL::R3;
    static constexpr int OFF2 = 0;
    static constexpr int OFF3 = OFF2 + pass_table_size<TEAM, PL::R2, NS2, M>();
    static constexpr int OFF4 = OFF3 + pass_table_size<TEAM, PL::R3, NS3, M>();
    static constexpr int OFFP = OFF4 + (PL::R4 ? pass_table_size<TEAM, (PL::R4 ? PL::R4 : 2), NS4, M>() : 0);
    static constexpr int SLOTS = OFFP + 2 * (M / 2 + 1);
};
template <typename T, typename PL> __host__ __device__ constexpr int pers_table_bytes() {
    return ((PersLayout<PL>::SLOTS * (int)sizeof(cx<T>) + 15) / 16) * 16;
}

template <typename T, typename PL>
__device__ __forceinline__ void pers_load_tables(unsigned char* smem, const cx<T>* __restrict__ blob) {
    cx<T>* l = reinterpret_cast<cx<T>*>(smem);
    for (int i = threadIdx.x; i < PersLayout<PL>::SLOTS; i += blockDim.x) l[i] = blob[i];
}

// forward / inverse transform of one channel-frame with every table in LDS
template <typename T, typename PL, bool INV>
__device__ __forceinline__ void fft_team_lt(cx<T>* buf, int t, const cx<T>* ltab) {
    constexpr int M = 1 << PL::LOG2M, TEAM = PL::TEAM, SH = PL::SH;
    using L = PersLayout<PL>;
    fft_pass_lt<T, M, TEAM, PL::R1, 1, INV, SH>(buf, t, ltab);
    fft_pass_lt<T, M, TEAM, PL::R2, L::NS2, INV, SH>(buf, t, ltab + L::OFF2);
    fft_pass_lt<T, M, TEAM, PL::R3, L::NS3, INV, SH>(buf, t, ltab + L::OFF3);
    if constexpr (PL::R4 != 0) fft_pass_lt<T, M, TEAM, PL::R4, L::NS4, INV, SH>(buf, t, ltab + L::OFF4);
}

// Last radix-4 pass of plan A (M = 1024) fused with the DCT pair step.  A lane takes the four
// butterflies k0 in {l, 64+l, 192-l, 256-l} (lane 0: {0, 64, 192, 128}); their outputs Z[k0 + 256 j] then
// contain both members of every pair (k, M-k) the DCT step needs, so that step runs on registers and the
// pass's own LDS write + read-back (and one team sync) disappear.  Writes X[0..N) over the buffer.
template <typename T, typename PL>
__device__ __forceinline__ void fft_last_pass_dct(cx<T>* buf, int l, const cx<T>* ltab, const cx<T>* lpost, T extra) {
    constexpr int M = 1 << PL::LOG2M, N = 2 * M, SH = PL::SH, TEAM = PL::TEAM;
    static_assert(M == 1024 && TEAM == 64 && PL::R3 == 4 && PL::R4 == 0, "plan A, N = 2048");
    using L = PersLayout<PL>;
    const cx<T>* tw3 = ltab + L::OFF3;
    constexpr int GOFF = M / 2 + 1;
    const bool lane0 = (l == 0);
    const int k0[4] = {l, 64 + l, 192 - l, lane0 ? 128 : 256 - l};
    cx<T> z[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) z[b][j] = buf[phys<T, SH>(k0[b] + 256 * j)];
    }
    team_sync<TEAM, true>();                               // every lane has its inputs: the buffer may be overwritten
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int nb = k0[b] >> 6, kk = k0[b] & 63;
#pragma unroll
        for (int j = 1; j < 4; ++j) z[b][j] = cmul(z[b][j], tw3[(nb * 3 + (j - 1)) * 64 + kk]);
        dft<4, false>(z[b]);
    }
    const T sc = ((T)1 / (T)(2 * N)) * extra, sc2 = (K<T>::s2 / (T)(2 * N)) * extra;
    auto pair = [&](int k, cx<T> zk, cx<T> zm) {            // zk = Z[k], zm = Z[M - k]
        const cx<T> zp = conj(zm);
        const cx<T> p = cmul(zk + zp, lpost[k]), q = cmul(zk - zp, lpost[k + GOFF]);
        const cx<T> S = p + q, D = p - q;
        real_slot<T, SH>(buf, k) = S.x * sc;
        if (k > 0) real_slot<T, SH>(buf, N - k) = -S.y * sc;
        if (k < M / 2) {
            real_slot<T, SH>(buf, M - k) = (D.x - D.y) * sc2;
            if (k > 0) real_slot<T, SH>(buf, M + k) = (D.x + D.y) * sc2;
        }
    };
    auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
    // pairs inside {64+l, 192-l}: (64 + l + 256 j) with (192 - l + 256 (3 - j)); the smaller index leads
    pair(64 + l, z[1][0], z[2][3]);
    pair(320 + l, z[1][1], z[2][2]);
    pair(448 - l, z[2][1], z[1][2]);
    pair(192 - l, z[2][0], z[1][3]);
    // pairs inside {l, 256-l}; lane 0 holds {0, 128} instead and pairs them with themselves:
    //   lane l >= 1: (l, 1024-l) (256+l, 768-l) (512-l, 512+l) (256-l, 768+l)
    //   lane 0     : (0, 0)      (256, 768)     (512, 512)     (128, 896)      + the extra pair (384, 640)
    pair(l, z[0][0], sel(z[0][0], z[3][3]));
    pair(256 + l, z[0][1], sel(z[0][3], z[3][2]));
    pair(512 - l, sel(z[0][2], z[3][1]), z[0][2]);
    pair(lane0 ? 128 : 256 - l, z[3][0], sel(z[3][3], z[0][3]));
    if (lane0) pair(384, z[3][1], z[3][2]);
    team_sync<TEAM, true>();
}

// Inverse counterpart (plan I: 4 16 16): the DCT pair step runs on registers and feeds the FIRST radix-4
// pass directly.  Lane l owns butterflies k0 in {l, 64+l, 192-l, 256-l} (lane 0: {0, 64, 192, 128}), i.e.
// exactly the 16 points Z'[k0 + 256 j] that its eight (k, M-k) pairs produce from X[k], X[N-k], X[M-k], X[M+k].
// Results go to slot k0 + 256 j (lane-contiguous); the middle pass reads them from there.
template <typename PL>
__device__ __forceinline__ void dct_pre_first_pass(cx<double>* buf, int l, const cx<double>* lpost) {
    using T = double;
    constexpr int M = 1 << PL::LOG2M, N = 2 * M, SH = PL::SH, TEAM = PL::TEAM;
    static_assert(M == 1024 && TEAM == 64 && PL::R1 == 4, "plan I, N = 2048");
    constexpr int GOFF = M / 2 + 1;
    const bool lane0 = (l == 0);
    struct PairOut { cx<T> zk, zm; };
    auto pair = [&](int k) -> PairOut {                      // -> Z'[k], Z'[M - k]
        const T xk = real_slot<T, SH>(buf, k);
        const T xnk = k > 0 ? real_slot<T, SH>(buf, N - k) : (T)0;
        const T a = real_slot<T, SH>(buf, M - k), b = real_slot<T, SH>(buf, k > 0 ? M + k : M);
        const cx<T> u = {xk, -xnk};
        const cx<T> s = {(a + b) * K<T>::s2, (b - a) * K<T>::s2};
        const cx<T> A = cmul(u + s, conj(lpost[k])), B = cmul(u - s, conj(lpost[k + GOFF]));
        return PairOut{A + B, conj(A - B)};
    };
    // group {64+l, 192-l}
    const PairOut p0 = pair(64 + l), p1 = pair(320 + l), p2 = pair(448 - l), p3 = pair(192 - l);
    // group {l, 256-l}; lane 0: {0, 128}
    const PairOut q0 = pair(l), q1 = pair(256 + l), q2 = pair(512 - l), q3 = pair(lane0 ? 128 : 256 - l);
    PairOut q4 = q3;
    if (lane0) q4 = pair(384);
    team_sync<TEAM, true>();                                 // all X read: the buffer may be overwritten
    auto sel = [&](cx<T> a, cx<T> b) { return cx<T>{lane0 ? a.x : b.x, lane0 ? a.y : b.y}; };
    cx<T> z[4][4];
    // butterfly 64+l : Z'[64+l], [320+l], [576+l] = M-(448-l), [832+l] = M-(192-l)
    z[1][0] = p0.zk; z[1][1] = p1.zk; z[1][2] = p2.zm; z[1][3] = p3.zm;
    // butterfly 192-l: Z'[192-l], [448-l], [704-l] = M-(320+l), [960-l] = M-(64+l)
    z[2][0] = p3.zk; z[2][1] = p2.zk; z[2][2] = p1.zm; z[2][3] = p0.zm;
    // butterfly l     : Z'[l], [256+l], [512+l] = M-(512-l), [768+l] = M-(256-l);  lane 0: Z'[0], [256], [512], [768] = M-256
    z[0][0] = q0.zk; z[0][1] = q1.zk; z[0][2] = sel(q2.zk, q2.zm); z[0][3] = sel(q1.zm, q3.zm);
    // butterfly 256-l : Z'[256-l], [512-l], [768-l] = M-(256+l), [1024-l] = M-l;  lane 0 (k0 = 128): Z'[128], [384], [640], [896]
    z[3][0] = q3.zk; z[3][1] = sel(q4.zk, q2.zk); z[3][2] = sel(q4.zm, q1.zm); z[3][3] = sel(q3.zm, q0.zm);
    const int k0[4] = {l, 64 + l, 192 - l, lane0 ? 128 : 256 - l};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        dft<4, true>(z[b]);
#pragma unroll
        for (int j = 0; j < 4; ++j) buf[phys<T, SH>(k0[b] + 256 * j)] = z[b][j];
    }
    team_sync<TEAM, true>();
}

// middle pass of plan I: radix 16, NS = 4, reading the first pass's results where it left them
// (butterfly b, output j at slot b + 256 j) and writing in Stockham order.
template <typename PL>
__device__ __forceinline__ void fft_middle_pass_inv(cx<double>* buf, int t, const cx<double>* ptab) {
    using T = double;
    constexpr int M = 1 << PL::LOG2M, SH = PL::SH, TEAM = PL::TEAM, R = 16, NS = 4;
    cx<T> v[R];
    const int k = t & 3;
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = buf[phys<T, SH>((t >> 2) + 16 * j + 256 * k)];     // Stockham index t + 64 j of pass 1
    team_sync<TEAM, true>();
#pragma unroll
    for (int j = 1; j < R; ++j) {
        cx<T> w = ptab[(j - 1) * NS + k];
        w.y = -w.y;
        v[j] = cmul(v[j], w);
    }
    dft<R, true>(v);
    const int base = (t - k) * R + k;
#pragma unroll
    for (int j = 0; j < R; ++j) buf[phys<T, SH>(base + j * NS)] = v[j];
    team_sync<TEAM, true>();
    (void)M;
}

// ---------------------------------------------------------------------------------------------
// encode.  grid = min(groups, CUs); block = teams * TEAM threads, teams = fpb * C (<= 8).
// Per iteration a thread stages CPT = N * itemsize / (16 * TEAM) 16-byte chunks of PCM.
// ---------------------------------------------------------------------------------------------
template <typename T, typename PL, int LG, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 2 : 1))
k_p0_fwd_pers(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<T>* __restrict__ blob, Geom g, int ngroups, int aligned_out) {
    constexpr int LOG2M = PL::LOG2M, M = 1 << LOG2M, N = 2 * M, TEAM = PL::TEAM, SH = PL::SH;
    constexpr int TB = pers_table_bytes<T, PL>();
    constexpr int CPT = (N << LG) / (16 * TEAM);
    constexpr int EPC = 16 >> LG;
    static_assert(CPT >= 1 && CPT * 16 * TEAM == (N << LG), "chunks per thread");
    FRAD_DYN_SMEM(smem);
    pers_load_tables<T, PL>(smem, blob);
    const cx<T>* ltab = reinterpret_cast<const cx<T>*>(smem);
    const cx<T>* lpost = ltab + PersLayout<PL>::OFFP;
    unsigned char* data = smem + TB;
    const int C = g.C, fpb = g.fpb, mode = g.in_mode;
    const int cf = threadIdx.x / TEAM, t0 = threadIdx.x - cf * TEAM;
    const long long rowb = (long long)C << LG;               // bytes per sample-frame
    const long long frameb = (g.frame_stride * C) << LG;     // bytes between frames
    const int cpf = (int)((N * rowb) / 16);                  // 16-byte chunks per frame
    const int slabs = (int)(rowb / 16);                      // mode 3
    uint32_t pf[CPT][4];
    int tid = threadIdx.x;                                   // refreshed (opaque) every iteration, see below

    // chunk index (inside the frame) of prefetch slot i, and its frame, for every stage mode
    auto locate = [&](int i, int& fl, int& ch) {
        if (mode == 3) {                                     // task = (quad, slab): 4 row chunks
            const int task = tid + (i >> 2) * blockDim.x;
            const int tpf = (N / 4) * slabs;
            fl = task / tpf;
            const int r = task - fl * tpf, zq = r / slabs, sl = r - zq * slabs;
            ch = (zq * 4 + (i & 3)) * slabs + sl;
        } else if (mode == 2) {                              // task = quad = 2 consecutive chunks
            const int task = tid + (i >> 1) * blockDim.x;
            fl = task / (N / 4);
            ch = (task - fl * (N / 4)) * 2 + (i & 1);
        } else {                                             // task = chunk
            const int task = tid + i * blockDim.x;
            fl = task / cpf; ch = task - fl * cpf;
        }
    };
    auto prefetch = [&](long long grp) {
        const long long f0 = grp * fpb;
        const long long rem = g.n_frames - f0;
        const int nfl = rem < fpb ? (int)rem : fpb;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {          // unconditional: a missing frame re-reads the last live one
            int fl, ch; locate(i, fl, ch);
            fl = fl < nfl ? fl : nfl - 1;
            load_words<4>(pcm + (f0 + fl) * frameb + (long long)ch * 16, pf[i]);
        }
    };
    auto put = [&](int fl, int c, int zq, T e0, T e1, T e2, T e3) {
        cx<T>* b = reinterpret_cast<cx<T>*>(data) + (long long)(fl * C + c) * M;
        b[phys<T, SH>(zq)] = cx<T>{e0, e2};
        b[phys<T, SH>(M - 1 - zq)] = cx<T>{e3, e1};
    };
    auto stage_write_c = [&](int nfl, auto code_tag, auto raw_tag) {
        constexpr int CODE = decltype(code_tag)::value;
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        if (mode == 3) {
            if constexpr (CPT % 4 == 0) {
#pragma unroll
                for (int k = 0; k < CPT / 4; ++k) {
                    const int task = tid + k * blockDim.x;
                    const int tpf = (N / 4) * slabs;
                    const int fl = task / tpf, r = task - fl * tpf, zq = r / slabs, sl = r - zq * slabs;
                    if (fl >= nfl) continue;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        T v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(pf[4 * k + i], e));
                        put(fl, sl * EPC + e, zq, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        } else if (mode == 2) {
            if constexpr (CPT % 2 == 0) {
                constexpr int CC = 8 >> LG;
#pragma unroll
                for (int k = 0; k < CPT / 2; ++k) {
                    const int task = tid + k * blockDim.x;
                    const int fl = task / (N / 4), zq = task - fl * (N / 4);
                    if (fl >= nfl) continue;
#pragma unroll
                    for (int c = 0; c < CC; ++c) {
                        T v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int idx = i * CC + c;                      // element inside the 32-byte quad
                            v[i] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(pf[2 * k + idx / EPC], idx % EPC));
                        }
                        put(fl, c, zq, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        } else {
            if constexpr (LG <= 2) {
                auto small = [&](auto lc_tag) {
                    constexpr int LC = decltype(lc_tag)::value, CC = 1 << LC, GPC = (EPC / CC) / 4;
                    if constexpr (GPC >= 1) {
#pragma unroll
                        for (int i = 0; i < CPT; ++i) {
                            const int task = tid + i * blockDim.x;
                            const int fl = task / cpf, ch = task - fl * cpf;
                            if (fl >= nfl) continue;
#pragma unroll
                            for (int gi = 0; gi < GPC; ++gi)
#pragma unroll
                                for (int c = 0; c < CC; ++c) {
                                    T v[4];
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = cvt_pcm_c<T, CODE, RAW>(word_elem<LG>(pf[i], (gi * 4 + r) * CC + c));
                                    put(fl, c, ch * GPC + gi, v[0], v[1], v[2], v[3]);
                                }
                        }
                    }
                };
                const int rb = (int)rowb;
                if (rb == 1) small(std::integral_constant<int, 0>{});
                else if (rb == 2) small(std::integral_constant<int, (LG <= 1 ? 1 - LG : 0)>{});
                else small(std::integral_constant<int, 2 - LG>{});
            }
        }
    };
    auto stage_write = [&](int nfl) { dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto c, auto r) { stage_write_c(nfl, c, r); }); };

    long long grp = blockIdx.x;
    if (grp < ngroups) prefetch(grp);
    __syncthreads();                                          // tables visible
    while (grp < ngroups) {
        const long long f0 = grp * fpb;
        const long long rem = g.n_frames - f0;
        const int nfl = rem < fpb ? (int)rem : fpb;
        tid = threadIdx.x; FRAD_OPAQUE(tid);                  // no hoisting of per-lane addresses out of the loop
        stage_write(nfl);
        FRAD_LDS_BARRIER();
        const long long next = grp + gridDim.x;
        if (next < ngroups) prefetch(next);                   // in flight during the butterflies
        int t = t0, cfo = cf * M;
        FRAD_OPAQUE(t); FRAD_OPAQUE(cfo);                     // recompute LDS addresses per iteration (no LICM)
        cx<T>* buf = reinterpret_cast<cx<T>*>(data) + cfo;
        fft_team_lt<T, PL, false>(buf, t, ltab);
        dct_post<T, LOG2M, 1, TEAM, SH, true>(buf, t, lpost);
        // retire the prefetch here (it had the whole transform to land; last iteration's stores are
        // long done too), so that the wait for it does not end up behind this iteration's stores
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) FRAD_OPAQUE(pf[i][j]);
        FRAD_LDS_BARRIER();
        if (g.cc_fast == 2) pack_out_pairs_any<T, SH, 2>(TB, payload, absmax, g, f0, nfl, M);
        else if (g.cc_fast == 1) pack_out_pairs_any<T, SH, 1>(TB, payload, absmax, g, f0, nfl, M);
        else pack_out_any<T, SH>(TB, payload, absmax, g, f0, nfl, M, aligned_out != 0);
        FRAD_LDS_BARRIER();
        grp = next;
    }
}

// ---------------------------------------------------------------------------------------------
// decode (always float64).  CC = channels (1 or 2), payload side in whole pack units.
// ---------------------------------------------------------------------------------------------
template <typename PL, int BITS, int CC, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 2 : 1))
k_p0_inv_pers(const unsigned char* __restrict__ payload, double* __restrict__ out,
              const cx<double>* __restrict__ blob, Geom g, int ngroups) {
    constexpr int LOG2M = PL::LOG2M, M = 1 << LOG2M, N = 2 * M, TEAM = PL::TEAM, SH = PL::SH;
    constexpr int TB = pers_table_bytes<double, PL>();
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int V = U > 2 * CC ? U : 2 * CC, KB = V / CC, UPV = V / U;
    constexpr int TPT = N / (V * TEAM);                       // tasks per thread and iteration
    static_assert(TPT >= 1 && TPT * V * TEAM == N, "tasks per thread");
    FRAD_DYN_SMEM(smem);
    pers_load_tables<double, PL>(smem, blob);
    const cx<double>* ltab = reinterpret_cast<const cx<double>*>(smem);
    const cx<double>* lpost = ltab + PersLayout<PL>::OFFP;
    unsigned char* data = smem + TB;
    const int fpb = g.fpb;
    const bool le = g.le && (BITS % 8 == 0);
    const int cf = threadIdx.x / TEAM, t0 = threadIdx.x - cf * TEAM;
    constexpr int tasks_pf = (N * CC) / V;                    // tasks per frame
    uint32_t pf[TPT][UPV][UB / 4];
    int tid = threadIdx.x;

    auto prefetch = [&](long long grp) {
        const long long f0 = grp * fpb;
        const long long rem = g.n_frames - f0;
        const int nfl = rem < fpb ? (int)rem : fpb;
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int task = tid + i * blockDim.x;
            int fl = task / tasks_pf;
            const int u = task - fl * tasks_pf;
            fl = fl < nfl ? fl : nfl - 1;                     // unconditional loads: no branch, no early wait
            const unsigned char* src = payload + (f0 + fl) * g.payload_stride;
#pragma unroll
            for (int w = 0; w < UPV; ++w) load_words<UB / 4>(src + ((long long)u * UPV + w) * UB, pf[i][w]);
        }
    };
    auto stage_write = [&](int nfl) {
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int task = tid + i * blockDim.x;
            const int fl = task / tasks_pf, u = task - fl * tasks_pf;
            if (fl >= nfl) continue;
            u64 codes[V];
#pragma unroll
            for (int w = 0; w < UPV; ++w) {
                u64 unit[U];
                unpack_unit<BITS>(pf[i][w], le, unit);
#pragma unroll
                for (int e = 0; e < U; ++e) codes[w * U + e] = unit[e];
            }
            const int s0 = (u * KB) >> 1;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                cx<double>* b = reinterpret_cast<cx<double>*>(data) + (long long)(fl * CC + c) * M;
#pragma unroll
                for (int kk = 0; kk < KB / 2; ++kk)
                    b[phys<double, SH>(s0 + kk)] = cx<double>{code_to_f64(codes[(2 * kk) * CC + c], BITS),
                                                              code_to_f64(codes[(2 * kk + 1) * CC + c], BITS)};
            }
        }
    };

    long long grp = blockIdx.x;
    if (grp < ngroups) prefetch(grp);
    __syncthreads();
    while (grp < ngroups) {
        const long long f0 = grp * fpb;
        const long long rem = g.n_frames - f0;
        const int nfl = rem < fpb ? (int)rem : fpb;
        tid = threadIdx.x; FRAD_OPAQUE(tid);
        stage_write(nfl);
        FRAD_LDS_BARRIER();
        const long long next = grp + gridDim.x;
        if (next < ngroups) prefetch(next);
        int t = t0, cfo = cf * M;
        FRAD_OPAQUE(t); FRAD_OPAQUE(cfo);
        cx<double>* buf = reinterpret_cast<cx<double>*>(data) + cfo;
        dct_pre_inverse<double, LOG2M, 1, TEAM, SH, true>(buf, t, lpost);
        fft_team_lt<double, PL, true>(buf, t, ltab);
#pragma unroll
        for (int i = 0; i < TPT; ++i)
#pragma unroll
            for (int w = 0; w < UPV; ++w)
#pragma unroll
                for (int j = 0; j < UB / 4; ++j) FRAD_OPAQUE(pf[i][w][j]);
        FRAD_LDS_BARRIER();
        store_pcm_quads<SH, CC>(TB, out, g, f0, nfl, M);
        FRAD_LDS_BARRIER();
        grp = next;
    }
}


// =============================================================================================
// Unit-synchronised persistent kernels (C = 1 or 2, plan A).
//
// The block-barrier kernels above put all eight waves of a CU in the same phase at the same time:
// LDS-write-bound staging, then VALU-bound butterflies, then the store burst -- each phase leaves
// the other pipes idle.  Here the waves of ONE FRAME (CC waves = its channels) form a unit that
// loops over frames on its own and synchronises only with itself through a tiny LDS counter
// barrier (the hardware s_barrier is block-wide).  The 8 / CC units of a CU drift apart, so one
// unit's staging overlaps another's butterflies and a third one's stores, while the 32 KiB of
// tables in LDS are still shared by all of them.
// =============================================================================================
#ifndef FRAD_HOST_EMULATION
#define FRAD_LDS_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FRAD_LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FRAD_NAP() __builtin_amdgcn_s_sleep(1)
#define FRAD_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FRAD_LDS_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define FRAD_LDS_LOAD(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define FRAD_NAP() std::this_thread::yield()
#define FRAD_LGKM0() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#endif

// barrier over the NW waves of a unit; `ctr` is the unit's LDS word, `epoch` a per-thread count
template <int NW> __device__ __forceinline__ void unit_barrier(unsigned* ctr, unsigned& epoch) {
    if constexpr (NW == 1) {
        team_sync<64>();
    } else {
        epoch += NW;
        __builtin_amdgcn_wave_barrier();                     // (lockstep on hardware; orders the emulator's lanes)
        FRAD_LGKM0();                                        // this wave's LDS traffic is done
        if ((threadIdx.x & 63) == 0) FRAD_LDS_ADD(ctr, 1u);
        while (FRAD_LDS_LOAD(ctr) < epoch) FRAD_NAP();       // every arrival is a +1; never reset
        FRAD_LGKM0();
    }
}

template <typename T, int BITS, int SH, int CC, int M>
__device__ __forceinline__ u64 pack_frame_pairs(const unsigned char* data, unsigned char* __restrict__ dst, bool le, int utid) {
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int V = U > 2 * CC ? U : 2 * CC, KB = V / CC;
    constexpr int tasks = (2 * M * CC) / V, uth = CC * 64, ITER = tasks / uth;
    if constexpr (ITER * uth != tasks) return 0;             // a 12-bit unit is wider than a lane's share at N = 1024: the host never selects it
    double fm = 0.0;
    bool nan = false;
    // compile-time trip count: the LDS reads of several tasks are in flight before the first conversion
#pragma unroll (ITER > 8 ? 8 : ITER)
    for (int it = 0; it < ITER; ++it) {
        const int u = utid + it * uth;
        u64 codes[V];
        const int s0 = (u * KB) >> 1;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const cx<T>* buf = reinterpret_cast<const cx<T>*>(data) + (long long)c * M;
#pragma unroll
            for (int kk = 0; kk < KB / 2; ++kk) {
                const cx<T> z = buf[phys<T, SH>(s0 + kk)];
                // |x| max as two float ops; NaN is tracked apart (fmax drops it) and re-imposed at the end
                fm = fmax(fm, fmax(fabs((double)z.x), fabs((double)z.y)));
                nan |= (z.x != z.x) | (z.y != z.y);
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
    return nan ? 0x7ff8000000000000ULL : d2u(fm);             // np.max(np.abs(.)) propagates NaN
}
// out-of-line so that the six storage formats do not count against the transform's registers
template <typename T, int SH, int CC, int M>
__device__ FRAD_NOINLINE void pack_frame_pairs_any(int data_off, unsigned char* __restrict__ dst, int wmax_off, int bits,
                                                   int le, int utid) {
    FRAD_DYN_SMEM(smem);
    const unsigned char* data = smem + data_off;
    u64 mx;
    switch (bits) {
        case 12: mx = pack_frame_pairs<T, 12, SH, CC, M>(data, dst, false, utid); break;
        case 16: mx = pack_frame_pairs<T, 16, SH, CC, M>(data, dst, le != 0, utid); break;
        case 24: mx = pack_frame_pairs<T, 24, SH, CC, M>(data, dst, le != 0, utid); break;
        case 32: mx = pack_frame_pairs<T, 32, SH, CC, M>(data, dst, le != 0, utid); break;
        case 48: mx = pack_frame_pairs<T, 48, SH, CC, M>(data, dst, le != 0, utid); break;
        default: mx = pack_frame_pairs<T, 64, SH, CC, M>(data, dst, le != 0, utid); break;
    }
    mx = wave_max_u64(mx);
    if ((threadIdx.x & 63) == 0) reinterpret_cast<u64*>(smem + wmax_off)[threadIdx.x >> 6] = mx;   // per-wave max, combined by the unit
}

template <typename T, typename PL, int LG, int CC>
__global__ void __launch_bounds__(512, UnitWaves<PL>::value)
k_p0_fwd_unit(const unsigned char* __restrict__ pcm, unsigned char* __restrict__ payload, double* absmax,
              const cx<T>* __restrict__ blob, Geom g) {
    constexpr int LOG2M = PL::LOG2M, M = 1 << LOG2M, N = 2 * M, TEAM = PL::TEAM, SH = PL::SH;
    static_assert(TEAM == 64, "one wave per channel-frame");
    constexpr int TB = pers_table_bytes<T, PL>(), CTRB = 96;     // 8 barrier words + 8 per-wave |X| maxima (N = 1024: two blocks fit 160 KiB)
    constexpr int CPT = (N << LG) / (16 * TEAM), EPC = 16 >> LG;
    constexpr int UTH = CC * 64, UPB = 8 / CC;               // threads per unit, units per block
    FRAD_DYN_SMEM(smem);
    pers_load_tables<T, PL>(smem, blob);
    unsigned* ctrs = reinterpret_cast<unsigned*>(smem + TB);
    if (threadIdx.x < 8) ctrs[threadIdx.x] = 0;
    const cx<T>* ltab = reinterpret_cast<const cx<T>*>(smem);
    const cx<T>* lpost = ltab + PersLayout<PL>::OFFP;
    const int unit = threadIdx.x / UTH, utid0 = threadIdx.x - unit * UTH;
    const int data_off = TB + CTRB + unit * CC * M * (int)sizeof(cx<T>);
    unsigned char* data = smem + data_off;                   // this unit's CC channel buffers
    unsigned* ctr = ctrs + unit;
    unsigned epoch = 0;
    const int mode = g.in_mode;
    const long long rowb = (long long)CC << LG, frameb = (g.frame_stride * CC) << LG;
    constexpr int cpf = (N * CC << LG) / 16, slabs = (CC << LG) / 16 > 0 ? (CC << LG) / 16 : 1;
    uint32_t pf[CPT][4];
    int utid = utid0;

    auto locate = [&](int i) -> int {                        // chunk (inside the frame) of prefetch slot i
        if (mode == 3) {
            const int task = utid + (i >> 2) * UTH, zq = task / slabs, sl = task - zq * slabs;
            return (zq * 4 + (i & 3)) * slabs + sl;
        } else if (mode == 2) {
            return (utid + (i >> 1) * UTH) * 2 + (i & 1);
        }
        return utid + i * UTH;
    };
    auto prefetch = [&](long long f) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) load_words<4>(pcm + f * frameb + (long long)locate(i) * 16, pf[i]);
    };
    auto put = [&](int c, int zq, T e0, T e1, T e2, T e3) {
        cx<T>* b = reinterpret_cast<cx<T>*>(data) + (long long)c * M;
        b[phys<T, SH>(zq)] = cx<T>{e0, e2};
        b[phys<T, SH>(M - 1 - zq)] = cx<T>{e3, e1};
    };
    auto stage_write_c = [&](auto code_tag, auto raw_tag) {
        constexpr int CODE = decltype(code_tag)::value;
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        if (mode == 3) {
            if constexpr (CPT % 4 == 0) {
#pragma unroll
                for (int k = 0; k < CPT / 4; ++k) {
                    const int task = utid + k * UTH, zq = task / slabs, sl = task - zq * slabs;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        T v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = cvt_pcm_c<T, CODE, RAW, true>(word_elem<LG>(pf[4 * k + i], e));
                        put(sl * EPC + e, zq, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        } else if (mode == 2) {
            if constexpr (CPT % 2 == 0 && (8 >> LG) == CC) {
#pragma unroll
                for (int k = 0; k < CPT / 2; ++k) {
                    const int zq = utid + k * UTH;
#pragma unroll
                    for (int c = 0; c < CC; ++c) {
                        T v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int idx = i * CC + c;
                            v[i] = cvt_pcm_c<T, CODE, RAW, true>(word_elem<LG>(pf[2 * k + idx / EPC], idx % EPC));
                        }
                        put(c, zq, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        } else {
            constexpr int GPC = (EPC / CC) / 4;
            if constexpr (GPC >= 1) {
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int ch = utid + i * UTH;
#pragma unroll
                    for (int gi = 0; gi < GPC; ++gi)
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            T v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = cvt_pcm_c<T, CODE, RAW, true>(word_elem<LG>(pf[i], (gi * 4 + r) * CC + c));
                            put(c, ch * GPC + gi, v[0], v[1], v[2], v[3]);
                        }
                }
            }
        }
    };

    const T deferred = (T)pcm_deferred_scale(g.dtype, g.raw_be);   // signed PCM stays un-normalised until the DCT step
    const long long stride = (long long)gridDim.x * UPB;
    long long f = (long long)blockIdx.x * UPB + unit;
    if (f < g.n_frames) prefetch(f);
    __syncthreads();                                          // tables and counters are set up
    while (f < g.n_frames) {
        utid = utid0; FRAD_OPAQUE(utid);                      // no hoisting of per-lane addresses out of the loop
        dispatch_pcm<LG>(g.dtype, g.raw_be, [&](auto c, auto r) { stage_write_c(c, r); });
        unit_barrier<CC>(ctr, epoch);
        const long long next = f + stride;
        if (next < g.n_frames) prefetch(next);                // lands while this frame is transformed
        int t = utid & 63, co = (utid >> 6) * M;
        FRAD_OPAQUE(t); FRAD_OPAQUE(co);
        cx<T>* buf = reinterpret_cast<cx<T>*>(data) + co;
        fft_pass_lt<T, M, TEAM, PL::R1, 1, false, SH>(buf, t, ltab);
        fft_pass_lt<T, M, TEAM, PL::R2, PersLayout<PL>::NS2, false, SH>(buf, t, ltab + PersLayout<PL>::OFF2);
        if constexpr (M == 1024) {
            fft_last_pass_dct<T, PL>(buf, t, ltab, lpost, deferred);
        } else {                                              // other sizes: last pass and DCT pair step through LDS
            fft_pass_lt<T, M, TEAM, PL::R3, PersLayout<PL>::NS3, false, SH>(buf, t, ltab + PersLayout<PL>::OFF3);
            dct_post<T, LOG2M, 1, TEAM, SH, true>(buf, t, lpost, deferred);
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) FRAD_OPAQUE(pf[i][j]);    // retire the prefetch before the store burst
        unit_barrier<CC>(ctr, epoch);
        pack_frame_pairs_any<T, SH, CC, M>(data_off, payload + f * g.payload_stride, TB + 32, g.bits, g.le, utid);
        unit_barrier<CC>(ctr, epoch);
        if (absmax != nullptr && utid0 == 0) {                // plain store: no atomics, no memset before the launch
            const u64* wm = reinterpret_cast<const u64*>(smem + TB + 32) + unit * CC;
            u64 m = wm[0];
            if constexpr (CC == 2) m = wm[1] > m ? wm[1] : m;
            *FRAD_GPTR(u64, absmax + f) = m;
        }
        f = next;
    }
}

// Decode epilogue of a unit: one interleaved output row (or row pair for mono) per lane, so that every
// store instruction of a wave writes 64 x 16 contiguous bytes.  Row n of channel c is component
// (n&3 in {1,2}) of slot z[n>>2] (n even) or z[M-1-(n>>2)] (n odd) -- Makhoul's permutation undone on the
// read side with conflict-free 8-byte LDS reads.
template <int SH, int CC, int M>
__device__ FRAD_NOINLINE void store_frame_rows(int data_off, double* __restrict__ dstf, int utid) {
    FRAD_DYN_SMEM(smem);
    const double* data = reinterpret_cast<const double*>(smem + data_off);
    constexpr int N = 2 * M, uth = CC * 64;
    auto sample = [&](int c, int n) -> double {
        const int q = n >> 2, r = n & 3;
        const int slot = (r & 1) ? (M - 1 - q) : q;
        return data[((long long)c * M + phys<double, SH>(slot)) * 2 + ((r == 1) | (r == 2))];
    };
    // compile-time trip counts: several rows' LDS reads are in flight before the first store
    if constexpr (CC == 2) {
#pragma unroll 8
        for (int i = 0; i < N / uth; ++i) {
            const int n = utid + i * uth;
            v2d v = {sample(0, n), sample(1, n)};
            FRAD_NT_STORE(v, FRAD_GPTR(v2d, dstf) + n);
        }
    } else {
#pragma unroll 8
        for (int i = 0; i < N / 2 / uth; ++i) {
            const int p = utid + i * uth;
            v2d v = {sample(0, 2 * p), sample(0, 2 * p + 1)};
            FRAD_NT_STORE(v, FRAD_GPTR(v2d, dstf) + p);
        }
    }
}

// The same rows leaving in the caller's PCM format (frad_p0_digital_pcm: backend/pcmformat.py:49-62 applied in the store; Geom::dtype
// names the format): a lane converts its row's CC samples, so a wave still writes one contiguous stretch per instruction.
template <int SH, int CC, int M>
__device__ FRAD_NOINLINE void store_frame_rows_pcm(int data_off, unsigned char* __restrict__ dstf, int utid, int dtype, int raw_be) {
    FRAD_DYN_SMEM(smem);
    const double* data = reinterpret_cast<const double*>(smem + data_off);
    constexpr int N = 2 * M, uth = CC * 64;
    auto sample = [&](int c, int n) -> double {
        const int q = n >> 2, r = n & 3;
        const int slot = (r & 1) ? (M - 1 - q) : q;
        return data[((long long)c * M + phys<double, SH>(slot)) * 2 + ((r == 1) | (r == 2))];
    };
    dispatch_out_format(dtype, [&](auto kind_tag, auto lg_tag) {
        constexpr int KIND = decltype(kind_tag)::value, LGS = decltype(lg_tag)::value;
        const bool be = (dtype & 1) != 0, raw = raw_be != 0 && be;
#pragma unroll 4
        for (int i = 0; i < N / uth; ++i) {
            const int n = utid + i * uth;
#pragma unroll
            for (int c = 0; c < CC; ++c)
                store_pcm_elem<LGS>(dstf + (((long long)n * CC + c) << LGS), from_f64_bits<KIND, LGS>(sample(c, n), raw), be);
        }
    });
}

template <typename PL, int BITS, int CC>
__global__ void __launch_bounds__(512, UnitWaves<PL>::value)
k_p0_inv_unit(const unsigned char* __restrict__ payload, double* __restrict__ out, const cx<double>* __restrict__ blob, Geom g) {
    constexpr int LOG2M = PL::LOG2M, M = 1 << LOG2M, N = 2 * M, TEAM = PL::TEAM, SH = PL::SH;
    static_assert(TEAM == 64, "one wave per channel-frame");
    constexpr int TB = pers_table_bytes<double, PL>(), CTRB = 32;      // 8 unit-barrier words (plan I fills the 160 KiB exactly)
    constexpr int U = unit_values(BITS), UB = unit_bytes(BITS);
    constexpr int V = U > 2 * CC ? U : 2 * CC, KB = V / CC, UPV = V / U;
    constexpr int TPT = N / (V * TEAM);
    static_assert(TPT >= 1 && TPT * V * TEAM == N, "tasks per thread");
    constexpr int UTH = CC * 64, UPB = 8 / CC;
    FRAD_DYN_SMEM(smem);
    pers_load_tables<double, PL>(smem, blob);
    unsigned* ctrs = reinterpret_cast<unsigned*>(smem + TB);
    if (threadIdx.x < 8) ctrs[threadIdx.x] = 0;
    const cx<double>* ltab = reinterpret_cast<const cx<double>*>(smem);
    const cx<double>* lpost = ltab + PersLayout<PL>::OFFP;
    const int unit = threadIdx.x / UTH, utid0 = threadIdx.x - unit * UTH;
    const int data_off = TB + CTRB + unit * CC * M * 16;
    unsigned char* data = smem + data_off;
    unsigned* ctr = ctrs + unit;
    unsigned epoch = 0;
    const bool le = g.le && (BITS % 8 == 0);
    uint32_t pf[TPT][UPV][UB / 4];
    int utid = utid0;

    auto prefetch = [&](long long f) {
        const unsigned char* src = payload + f * g.payload_stride;
#pragma unroll
        for (int i = 0; i < TPT; ++i)
#pragma unroll
            for (int w = 0; w < UPV; ++w) load_words<UB / 4>(src + ((long long)(utid + i * UTH) * UPV + w) * UB, pf[i][w]);
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int u = utid + i * UTH;
            u64 codes[V];
#pragma unroll
            for (int w = 0; w < UPV; ++w) {
                u64 un[U];
                unpack_unit<BITS>(pf[i][w], le, un);
#pragma unroll
                for (int e = 0; e < U; ++e) codes[w * U + e] = un[e];
            }
            const int s0 = (u * KB) >> 1;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                cx<double>* b = reinterpret_cast<cx<double>*>(data) + (long long)c * M;
#pragma unroll
                for (int kk = 0; kk < KB / 2; ++kk)
                    b[phys<double, SH>(s0 + kk)] = cx<double>{code_to_f64(codes[(2 * kk) * CC + c], BITS),
                                                              code_to_f64(codes[(2 * kk + 1) * CC + c], BITS)};
            }
        }
    };

    const long long stride = (long long)gridDim.x * UPB;
    long long f = (long long)blockIdx.x * UPB + unit;
    if (f < g.n_frames) prefetch(f);
    __syncthreads();
    while (f < g.n_frames) {
        utid = utid0; FRAD_OPAQUE(utid);
        stage_write();
        unit_barrier<CC>(ctr, epoch);
        const long long next = f + stride;
        if (next < g.n_frames) prefetch(next);
        int t = utid & 63, co = (utid >> 6) * M;
        FRAD_OPAQUE(t); FRAD_OPAQUE(co);
        cx<double>* buf = reinterpret_cast<cx<double>*>(data) + co;
        if constexpr (PL::R1 == 4) {                          // plan I: pair step + first pass on registers
            dct_pre_first_pass<PL>(buf, t, lpost);
            fft_middle_pass_inv<PL>(buf, t, ltab + PersLayout<PL>::OFF2);
            fft_pass_lt<double, M, TEAM, PL::R3, PersLayout<PL>::NS3, true, SH>(buf, t, ltab + PersLayout<PL>::OFF3);
        } else {
            dct_pre_inverse<double, LOG2M, 1, TEAM, SH, true>(buf, t, lpost);
            fft_team_lt<double, PL, true>(buf, t, ltab);
        }
#pragma unroll
        for (int i = 0; i < TPT; ++i)
#pragma unroll
            for (int w = 0; w < UPV; ++w)
#pragma unroll
                for (int j = 0; j < UB / 4; ++j) FRAD_OPAQUE(pf[i][w][j]);
        unit_barrier<CC>(ctr, epoch);
        if (g.dtype != 22) {                                  // (uniform) the caller's PCM format instead of float64
            const int lgs = (g.dtype >> 1) & 3;
            store_frame_rows_pcm<SH, CC, M>(data_off, reinterpret_cast<unsigned char*>(out) + ((f * (long long)N * CC) << lgs), utid, g.dtype, g.raw_be);
        } else
        store_frame_rows<SH, CC, M>(data_off, out + f * (long long)N * CC, utid);
        unit_barrier<CC>(ctr, epoch);
        f = next;
    }
}

}  // namespace frad
