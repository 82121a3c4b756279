// persistent profile-0 kernels (frad_persistent.hpp): instantiation + launch policy
#include "frad_persistent.hpp"
#include "frad_launch.hpp"
#include <cstdlib>
#include <vector>

namespace frad {
namespace {

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
bool disabled() { const char* e = tune("FRAD_TUNE_NO_PERS"); return e && e[0] == '1'; }
int blocks_per_cu() { const char* e = tune("FRAD_TUNE_PERS_BPC"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 4 ? v : 1; }

template <typename T, typename PL, int LG, int MAXT>
void go_fwd(const void* blob, int threads, size_t lds, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
            double* am, const Geom& g, int ngroups, int ao) {
    allow_lds(k_p0_fwd_pers<T, PL, LG, MAXT>, lds);
    hipLaunchKernelGGL((k_p0_fwd_pers<T, PL, LG, MAXT>), dim3(grid), dim3(threads), lds, s, pcm, pay, am,
                       static_cast<const cx<T>*>(blob), g, ngroups, ao);
}

template <typename PL, int MAXT, int BITS>
void go_inv(const void* blobv, int cc, int threads, size_t lds, int grid, hipStream_t s, const unsigned char* pay, double* out,
            const Geom& g, int ngroups) {
    const cx<double>* blob = static_cast<const cx<double>*>(blobv);
    if constexpr (unit_values(BITS) * PL::TEAM > (2 << PL::LOG2M)) {
        (void)blob; (void)cc; (void)threads; (void)lds; (void)grid; (void)s; (void)pay; (void)out; (void)g; (void)ngroups;
    } else
    if (cc == 2) {
        allow_lds(k_p0_inv_pers<PL, BITS, 2, MAXT>, lds);
        hipLaunchKernelGGL((k_p0_inv_pers<PL, BITS, 2, MAXT>), dim3(grid), dim3(threads), lds, s, pay, out, blob, g, ngroups);
    } else {
        allow_lds(k_p0_inv_pers<PL, BITS, 1, MAXT>, lds);
        hipLaunchKernelGGL((k_p0_inv_pers<PL, BITS, 1, MAXT>), dim3(grid), dim3(threads), lds, s, pay, out, blob, g, ngroups);
    }
}
template <typename PL, int MAXT>
void go_inv_bits(const void* blob, int cc, int threads, size_t lds, int grid, hipStream_t s, const unsigned char* pay, double* out,
                 const Geom& g, int ngroups) {
    switch (g.bits) {
        case 12: go_inv<PL, MAXT, 12>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
        case 16: go_inv<PL, MAXT, 16>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
        case 24: go_inv<PL, MAXT, 24>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
        case 32: go_inv<PL, MAXT, 32>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
        case 48: go_inv<PL, MAXT, 48>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
        default: go_inv<PL, MAXT, 64>(blob, cc, threads, lds, grid, s, pay, out, g, ngroups); break;
    }
}
template <int LG, typename PL = PlanA10>
void go_fwd_unit(const void* blob, int cc, size_t lds, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                 double* am, const Geom& g) {
    const cx<double>* b = static_cast<const cx<double>*>(blob);
    if (cc == 2) {
        allow_lds(k_p0_fwd_unit<double, PL, LG, 2>, lds);
        hipLaunchKernelGGL((k_p0_fwd_unit<double, PL, LG, 2>), dim3(grid), dim3(512), lds, s, pcm, pay, am, b, g);
    } else {
        allow_lds(k_p0_fwd_unit<double, PL, LG, 1>, lds);
        hipLaunchKernelGGL((k_p0_fwd_unit<double, PL, LG, 1>), dim3(grid), dim3(512), lds, s, pcm, pay, am, b, g);
    }
}
template <int BITS, typename PL = PlanA10>
void go_inv_unit_a(const void* blob, int cc, size_t lds, int grid, hipStream_t s, const unsigned char* pay, double* out, const Geom& g) {
    const cx<double>* b = static_cast<const cx<double>*>(blob);
    if (cc == 2) {
        allow_lds(k_p0_inv_unit<PL, BITS, 2>, lds);
        hipLaunchKernelGGL((k_p0_inv_unit<PL, BITS, 2>), dim3(grid), dim3(512), lds, s, pay, out, b, g);
    } else {
        allow_lds(k_p0_inv_unit<PL, BITS, 1>, lds);
        hipLaunchKernelGGL((k_p0_inv_unit<PL, BITS, 1>), dim3(grid), dim3(512), lds, s, pay, out, b, g);
    }
}
// default: the unfused 16-16-4 inverse (plan A).  FRAD_TUNE_INV_PLAN=I selects the 4-16-16 plan with the pair step
// fused into the first pass; measured equal within noise on MI355X (decode already runs at ~80 % of the achievable
// HBM rate), and plan A leaves a little LDS headroom.
bool inv_plan_a() { const char* e = tune("FRAD_TUNE_INV_PLAN"); return !(e && (e[0] == 'I' || e[0] == 'i')); }

template <int BITS>
void go_inv_unit(const void* blob, int cc, size_t lds, int grid, hipStream_t s, const unsigned char* pay, double* out, const Geom& g) {
    const cx<double>* b = static_cast<const cx<double>*>(blob);
    if (cc == 2) {
        allow_lds(k_p0_inv_unit<PlanI10, BITS, 2>, lds);
        hipLaunchKernelGGL((k_p0_inv_unit<PlanI10, BITS, 2>), dim3(grid), dim3(512), lds, s, pay, out, b, g);
    } else {
        allow_lds(k_p0_inv_unit<PlanI10, BITS, 1>, lds);
        hipLaunchKernelGGL((k_p0_inv_unit<PlanI10, BITS, 1>), dim3(grid), dim3(512), lds, s, pay, out, b, g);
    }
}
bool unit_sync() { const char* e = tune("FRAD_TUNE_PERS_UNIT"); return !(e && e[0] == '0'); }

// plan of the N = 2048 float64 kernels: B = two waves per channel-frame (4 waves/SIMD), A = one
bool plan_b() { const char* e = tune("FRAD_TUNE_PERS_PLAN"); return e && (e[0] == 'B' || e[0] == 'b'); }

}  // namespace

// Host image of the LDS table blob (PersLayout<PL>): pass tables in lane-linear order, then w_k, g_k.
// `unit(p, q, re, im)` must return exp(-i pi p / q).
template <typename T, typename PL>
static void fill_blob(std::vector<unsigned char>& bytes, void (*unit)(long long, long long, long double&, long double&)) {
    using L = PersLayout<PL>;
    constexpr int M = 1 << PL::LOG2M, N = 2 * M, TEAM = PL::TEAM;
    std::vector<cx<T>> out(L::SLOTS, cx<T>{0, 0});
    auto pass = [&](int off, int R, int NS) {
        if (R == 0) return;
        const int NB = M / R / TEAM, KW = NS > TEAM ? TEAM : NS, nbn = NS > TEAM ? NB : 1;
        for (int nb = 0; nb < nbn; ++nb)
            for (int j = 1; j < R; ++j)
                for (int kk = 0; kk < KW; ++kk) {
                    const long long k = (kk + (long long)nb * TEAM) & (NS - 1);
                    long double re, im; unit(2 * j * k, (long long)NS * R, re, im);      // W_{NS R}^{j k}
                    out[off + (nb * (R - 1) + (j - 1)) * KW + kk] = cx<T>{(T)re, (T)im};
                }
    };
    pass(L::OFF2, PL::R2, L::NS2);
    pass(L::OFF3, PL::R3, L::NS3);
    pass(L::OFF4, PL::R4, L::NS4);
    for (int k = 0; k <= M / 2; ++k) {
        long double re, im;
        unit(k, 2LL * N, re, im); out[L::OFFP + k] = cx<T>{(T)re, (T)im};
        unit((long long)N + 5LL * k, 2LL * N, re, im); out[L::OFFP + M / 2 + 1 + k] = cx<T>{(T)re, (T)im};
    }
    bytes.assign((unsigned char*)out.data(), (unsigned char*)(out.data() + out.size()));
}
// blob `which` (0 = plan A, 1 = plan B) for (log2m, f32); returns its size, 0 when there is none
size_t pers_blob_build(int log2m, bool f32, int which, std::vector<unsigned char>& bytes,
                       void (*unit)(long long, long long, long double&, long double&)) {
    bytes.clear();
    if (log2m == 10 && !f32) {
        if (which == 0) fill_blob<double, PlanA10>(bytes, unit);
        else if (which == 1) fill_blob<double, PlanB10>(bytes, unit);
        else fill_blob<double, PlanI10>(bytes, unit);
    }
    else if (log2m == 11 && f32 && which == 0) fill_blob<float, PlanA11>(bytes, unit);
    else if (log2m == 9 && !f32 && which == 0) fill_blob<double, PlanA9>(bytes, unit);
    return bytes.size();
}

// returns 1 when the persistent kernel took the launch, 0 when the geometry is not one of its own
int launch_p0_fwd_pers(bool f32, int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                       double* am, const Tables& tb, Geom g, int ao) {
    if (disabled() || tb.blob == nullptr || c.cg != g.C || g.in_mode == 0 || g.n_valid != g.N || g.C > 8) return 0;
    const bool n1024 = !f32 && c.log2m == 9 && lg >= 1 && lg <= 3 && g.bits != 12 && !tune("FRAD_TUNE_NO_UNIT9");
    const bool geom_ok = f32 ? (c.log2m == 11 && lg == 2) : ((c.log2m == 10 && lg >= 1 && lg <= 3) || n1024);
    if (!geom_ok) return 0;
    if (!f32 && unit_sync() && !plan_b() && g.C <= 2 && g.cc_fast == g.C &&
        ((g.in_mode == 1) || (g.in_mode == 2 && (8 >> lg) == g.C) || (g.in_mode == 3 && lg >= (g.C == 1 ? 4 : 3)))) {
        // unit-synchronised kernel: one frame per unit of C waves, 8 / C units per block
        const int upb = 8 / g.C;
        const long long nb = (g.n_frames + upb - 1) / upb;
        if (n1024) {                                          // 8 KiB per channel-frame: two blocks per CU
            const size_t lds = (size_t)pers_table_bytes<double, PlanA9>() + 96 + 8 * 512 * 16;     // 2 x 81 920 B = the whole LDS
            const long long cap = (long long)cu_count() * 2;
            const int grid = (int)(nb < cap ? nb : cap);
            if (lg == 1) go_fwd_unit<1, PlanA9>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
            else if (lg == 2) go_fwd_unit<2, PlanA9>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
            else go_fwd_unit<3, PlanA9>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
            return 1;
        }
        const size_t lds = (size_t)pers_table_bytes<double, PlanA10>() + 96 + 8 * 1024 * 16;
        const long long cap = (long long)cu_count() * blocks_per_cu();
        const int grid = (int)(nb < cap ? nb : cap);
        if (lg == 1) go_fwd_unit<1>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
        else if (lg == 2) go_fwd_unit<2>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
        else go_fwd_unit<3>(tb.blob, g.C, lds, grid, s, pcm, pay, am, g);
        return 1;
    }
    if (n1024) return 0;                                      // the block-barrier variant is not built for N = 1024
    if (am != nullptr && hipMemsetAsync(am, 0, sizeof(double) * (size_t)g.n_frames, s) != hipSuccess) return 0;
    const bool pb = !f32 && plan_b() && tb.blob_b != nullptr && lg <= 2;
    const int team = f32 ? PlanA11::TEAM : pb ? PlanB10::TEAM : PlanA10::TEAM, M = 1 << c.log2m;
    const int cpt = (int)(((long long)g.N << lg) / (16 * team));
    if (cpt < 1 || (g.in_mode == 2 && cpt % 2) || (g.in_mode == 3 && cpt % 4)) return 0;
    const int teams = (8 / g.C) * g.C;
    g.fpb = teams / g.C;
    const int threads = teams * team;
    const size_t tbytes = f32 ? pers_table_bytes<float, PlanA11>() : pb ? pers_table_bytes<double, PlanB10>() : pers_table_bytes<double, PlanA10>();
    const size_t lds = tbytes + (size_t)teams * M * (f32 ? 8 : 16);
    const long long ngroups = (g.n_frames + g.fpb - 1) / g.fpb;
    if (ngroups > 0x7fffffffLL) return 0;
    const long long cap = (long long)cu_count() * blocks_per_cu();
    const int grid = (int)(ngroups < cap ? ngroups : cap);
    const int ng = (int)ngroups;
    if (f32) go_fwd<float, PlanA11, 2, 512>(tb.blob, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
    else if (pb) {
        if (lg == 1) go_fwd<double, PlanB10, 1, 1024>(tb.blob_b, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
        else go_fwd<double, PlanB10, 2, 1024>(tb.blob_b, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
    }
    else if (lg == 1) go_fwd<double, PlanA10, 1, 512>(tb.blob, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
    else if (lg == 2) go_fwd<double, PlanA10, 2, 512>(tb.blob, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
    else go_fwd<double, PlanA10, 3, 512>(tb.blob, threads, lds, grid, s, pcm, pay, am, g, ng, ao);
    return 1;
}

int launch_p0_inv_pers(const FastCfg& c, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb, Geom g) {
    if (disabled() || tb.blob == nullptr || c.cg != g.C || (c.log2m != 10 && c.log2m != 9) || g.C > 2 || g.cc_fast != g.C || g.in_mode != g.C) return 0;
    if (c.log2m == 9) {                                       // N = 1024: unit kernel only, two blocks per CU
        if (!unit_sync() || plan_b() || g.bits == 12 || tune("FRAD_TUNE_NO_UNIT9")) return 0;
        const int upb = 8 / g.C;
        const size_t lds = (size_t)pers_table_bytes<double, PlanA9>() + 32 + 8 * 512 * 16;
        const long long nb = (g.n_frames + upb - 1) / upb, cap = (long long)cu_count() * 2;
        const int grid = (int)(nb < cap ? nb : cap);
        switch (g.bits) {
            case 16: go_inv_unit_a<16, PlanA9>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 24: go_inv_unit_a<24, PlanA9>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 32: go_inv_unit_a<32, PlanA9>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 48: go_inv_unit_a<48, PlanA9>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            default: go_inv_unit_a<64, PlanA9>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
        }
        return 1;
    }
    if (unit_sync() && !plan_b() && inv_plan_a()) {
        const int upb = 8 / g.C;
        const size_t lds = (size_t)pers_table_bytes<double, PlanA10>() + 32 + 8 * 1024 * 16;
        const long long nb = (g.n_frames + upb - 1) / upb, cap = (long long)cu_count() * blocks_per_cu();
        const int grid = (int)(nb < cap ? nb : cap);
        switch (g.bits) {
            case 12: go_inv_unit_a<12>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 16: go_inv_unit_a<16>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 24: go_inv_unit_a<24>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 32: go_inv_unit_a<32>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            case 48: go_inv_unit_a<48>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
            default: go_inv_unit_a<64>(tb.blob, g.C, lds, grid, s, pay, out, g); break;
        }
        return 1;
    }
    if (unit_sync() && !plan_b() && tb.blob_i != nullptr) {
        const int upb = 8 / g.C;
        const size_t lds = (size_t)pers_table_bytes<double, PlanI10>() + 32 + 8 * 1024 * 16;   // = 163840 B, all of a CU's LDS
        const long long nb = (g.n_frames + upb - 1) / upb, cap = (long long)cu_count() * blocks_per_cu();
        const int grid = (int)(nb < cap ? nb : cap);
        switch (g.bits) {
            case 12: go_inv_unit<12>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
            case 16: go_inv_unit<16>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
            case 24: go_inv_unit<24>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
            case 32: go_inv_unit<32>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
            case 48: go_inv_unit<48>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
            default: go_inv_unit<64>(tb.blob_i, g.C, lds, grid, s, pay, out, g); break;
        }
        return 1;
    }
    if (g.dtype != 22) return 0;                              // (not FRAD_PCM_F64LE; only the unit kernels' store converts: frad_p0_digital_pcm)
    const bool pb = plan_b() && tb.blob_b != nullptr && g.bits != 12;   // a 12-bit unit is wider than plan B's lane share
    const int teams = 8, team = pb ? PlanB10::TEAM : PlanA10::TEAM;
    g.fpb = teams / g.C;
    const int threads = teams * team;
    const size_t lds = (size_t)(pb ? pers_table_bytes<double, PlanB10>() : pers_table_bytes<double, PlanA10>()) + (size_t)teams * 1024 * 16;
    const long long ngroups = (g.n_frames + g.fpb - 1) / g.fpb;
    if (ngroups > 0x7fffffffLL) return 0;
    const long long cap = (long long)cu_count() * blocks_per_cu();
    const int grid = (int)(ngroups < cap ? ngroups : cap);
    if (pb) go_inv_bits<PlanB10, 1024>(tb.blob_b, g.C, threads, lds, grid, s, pay, out, g, (int)ngroups);
    else go_inv_bits<PlanA10, 512>(tb.blob, g.C, threads, lds, grid, s, pay, out, g, (int)ngroups);
    return 1;
}

}  // namespace frad
