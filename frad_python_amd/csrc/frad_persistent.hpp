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

namespace frad {

// Plans of the persistent kernels: radices R1..R4 (0 = no such pass), lanes per channel-frame,
// swizzle shift (log2 R1).
struct PlanA10 { static constexpr int LOG2M = 10, TEAM = 64,  SH = 4, R1 = 16, R2 = 16, R3 = 4, R4 = 0; };  // 2 waves/SIMD
struct PlanB10 { static constexpr int LOG2M = 10, TEAM = 128, SH = 3, R1 = 8,  R2 = 8,  R3 = 8, R4 = 2; };  // 4 waves/SIMD
struct PlanA11 { static constexpr int LOG2M = 11, TEAM = 128, SH = 4, R1 = 16, R2 = 16, R3 = 8, R4 = 0; };

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

}  // namespace frad
