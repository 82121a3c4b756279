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
bool disabled() { const char* e = getenv("FRAD_TUNE_NO_PERS"); return e && e[0] == '1'; }
int blocks_per_cu() { const char* e = getenv("FRAD_TUNE_PERS_BPC"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 4 ? v : 1; }

template <typename T, int LOG2M, int LG, int MAXT>
void go_fwd(int threads, size_t lds, int grid, hipStream_t s, const unsigned char* pcm, unsigned char* pay, double* am,
            const Tables& tb, const Geom& g, int ngroups, int ao) {
    allow_lds(k_p0_fwd_pers<T, LOG2M, LG, MAXT>, lds);
    hipLaunchKernelGGL((k_p0_fwd_pers<T, LOG2M, LG, MAXT>), dim3(grid), dim3(threads), lds, s, pcm, pay, am,
                       static_cast<const cx<T>*>(tb.blob), g, ngroups, ao);
}

template <int BITS>
void go_inv(int cc, int threads, size_t lds, int grid, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb,
            const Geom& g, int ngroups) {
    const cx<double>* blob = static_cast<const cx<double>*>(tb.blob);
    if (cc == 2) {
        allow_lds(k_p0_inv_pers<10, BITS, 2, 512>, lds);
        hipLaunchKernelGGL((k_p0_inv_pers<10, BITS, 2, 512>), dim3(grid), dim3(threads), lds, s, pay, out, blob, g, ngroups);
    } else {
        allow_lds(k_p0_inv_pers<10, BITS, 1, 512>, lds);
        hipLaunchKernelGGL((k_p0_inv_pers<10, BITS, 1, 512>), dim3(grid), dim3(threads), lds, s, pay, out, blob, g, ngroups);
    }
}

}  // namespace

// Host image of the LDS table blob (PersLayout<LOG2M>): pass tables in lane-linear order, then w_k, g_k.
// `unit(p, q, re, im)` must return exp(-i pi p / q).
template <typename T, int LOG2M>
static void fill_blob(std::vector<cx<T>>& out, void (*unit)(long long, long long, long double&, long double&)) {
    using L = PersLayout<LOG2M>; using P = PersPlan<LOG2M>;
    constexpr int M = 1 << LOG2M, N = 2 * M, TEAM = Plan<LOG2M>::TEAM;
    out.assign(L::SLOTS, cx<T>{0, 0});
    auto pass = [&](int off, int R, int NS) {
        const int NB = M / R / TEAM, KW = NS > TEAM ? TEAM : NS, nbn = NS > TEAM ? NB : 1;
        for (int nb = 0; nb < nbn; ++nb)
            for (int j = 1; j < R; ++j)
                for (int kk = 0; kk < KW; ++kk) {
                    const long long k = (kk + (long long)nb * TEAM) & (NS - 1);
                    long double re, im; unit(2 * j * k, (long long)NS * R, re, im);      // W_{NS R}^{j k}
                    out[off + (nb * (R - 1) + (j - 1)) * KW + kk] = cx<T>{(T)re, (T)im};
                }
    };
    pass(L::OFF2, P::R2, P::NS2);
    pass(L::OFF3, P::R3, P::NS3);
    for (int k = 0; k <= M / 2; ++k) {
        long double re, im;
        unit(k, 2LL * N, re, im); out[L::OFFP + k] = cx<T>{(T)re, (T)im};
        unit((long long)N + 5LL * k, 2LL * N, re, im); out[L::OFFP + M / 2 + 1 + k] = cx<T>{(T)re, (T)im};
    }
}
// bytes of the blob for (log2m, f32), 0 when the plan has no persistent kernel
size_t pers_blob_build(int log2m, bool f32, std::vector<unsigned char>& bytes,
                       void (*unit)(long long, long long, long double&, long double&)) {
    bytes.clear();
    if (log2m == 10 && !f32) { std::vector<cx<double>> v; fill_blob<double, 10>(v, unit); bytes.assign((unsigned char*)v.data(), (unsigned char*)(v.data() + v.size())); }
    else if (log2m == 11 && f32) { std::vector<cx<float>> v; fill_blob<float, 11>(v, unit); bytes.assign((unsigned char*)v.data(), (unsigned char*)(v.data() + v.size())); }
    return bytes.size();
}

// returns 1 when the persistent kernel took the launch, 0 when the geometry is not one of its own
int launch_p0_fwd_pers(bool f32, int lg, const FastCfg& c, hipStream_t s, const unsigned char* pcm, unsigned char* pay,
                       double* am, const Tables& tb, Geom g, int ao) {
    if (disabled() || tb.blob == nullptr || c.cg != g.C || g.in_mode == 0 || g.n_valid != g.N || g.C > 8) return 0;
    const bool geom_ok = f32 ? (c.log2m == 11 && lg == 2) : (c.log2m == 10 && lg >= 1 && lg <= 3);
    if (!geom_ok) return 0;
    const int team = c.team, M = 1 << c.log2m;
    const int cpt = (int)(((long long)g.N << lg) / (16 * team));
    if ((g.in_mode == 2 && cpt % 2) || (g.in_mode == 3 && cpt % 4)) return 0;
    const int teams = (8 / g.C) * g.C;
    g.fpb = teams / g.C;
    const int threads = teams * team;
    const size_t lds = (size_t)(f32 ? pers_table_bytes<float, 11>() : pers_table_bytes<double, 10>()) +
                       (size_t)teams * M * (f32 ? 8 : 16);
    const long long ngroups = (g.n_frames + g.fpb - 1) / g.fpb;
    if (ngroups > 0x7fffffffLL) return 0;
    const long long cap = (long long)cu_count() * blocks_per_cu();
    const int grid = (int)(ngroups < cap ? ngroups : cap);
    if (f32) go_fwd<float, 11, 2, 1024>(threads, lds, grid, s, pcm, pay, am, tb, g, (int)ngroups, ao);
    else if (lg == 1) go_fwd<double, 10, 1, 512>(threads, lds, grid, s, pcm, pay, am, tb, g, (int)ngroups, ao);
    else if (lg == 2) go_fwd<double, 10, 2, 512>(threads, lds, grid, s, pcm, pay, am, tb, g, (int)ngroups, ao);
    else go_fwd<double, 10, 3, 512>(threads, lds, grid, s, pcm, pay, am, tb, g, (int)ngroups, ao);
    return 1;
}

int launch_p0_inv_pers(const FastCfg& c, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb, Geom g) {
    if (disabled() || tb.blob == nullptr || c.cg != g.C || c.log2m != 10 || g.C > 2 || g.cc_fast != g.C || g.in_mode != g.C) return 0;
    const int teams = 8;
    g.fpb = teams / g.C;
    const int threads = teams * c.team;
    const size_t lds = (size_t)pers_table_bytes<double, 10>() + (size_t)teams * 1024 * 16;
    const long long ngroups = (g.n_frames + g.fpb - 1) / g.fpb;
    if (ngroups > 0x7fffffffLL) return 0;
    const long long cap = (long long)cu_count() * blocks_per_cu();
    const int grid = (int)(ngroups < cap ? ngroups : cap);
    switch (g.bits) {
        case 12: go_inv<12>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
        case 16: go_inv<16>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
        case 24: go_inv<24>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
        case 32: go_inv<32>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
        case 48: go_inv<48>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
        default: go_inv<64>(g.C, threads, lds, grid, s, pay, out, tb, g, (int)ngroups); break;
    }
    return 1;
}

}  // namespace frad
