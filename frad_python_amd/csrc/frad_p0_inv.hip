// k_p0_inv (always float64), all sizes
#include "frad_launch.hpp"

namespace frad {

int launch_p0_inv(const FastCfg& c, dim3 grid, hipStream_t s, const unsigned char* pay, double* out, const Tables& tb,
                  const Geom& g, int ai) {
    const cx<double>* tw = static_cast<const cx<double>*>(tb.tw);
    const cx<double>* post = static_cast<const cx<double>*>(tb.post);
#define FRAD_GO(L, MAXT) do { allow_lds(k_p0_inv<L, MAXT>, c.lds); \
        hipLaunchKernelGGL((k_p0_inv<L, MAXT>), grid, dim3(c.threads), c.lds, s, pay, out, tw, post, g, ai); } while (0)
#define FRAD_GRPT(L, MAXT) do { allow_lds(k_p0_inv_grp<L, MAXT>, c.lds); \
        hipLaunchKernelGGL((k_p0_inv_grp<L, MAXT>), grid, dim3(c.threads), c.lds, s, pay, out, tw, post, g); } while (0)
#define FRAD_GRP(L) case L: if (c.threads <= 512) FRAD_GRPT(L, 512); else FRAD_GRPT(L, 1024); return 0;
    if (c.cg < g.C) {
        switch (c.log2m) { FRAD_GRP(8) FRAD_GRP(9) FRAD_GRP(10) FRAD_GRP(11) FRAD_GRP(12) FRAD_GRP(13) default: return -2; }
    }
#undef FRAD_GRP
#undef FRAD_GRPT
#define FRAD_CASE(L) case L: if (c.threads <= 256) FRAD_GO(L, 256); else if (c.threads <= 512) FRAD_GO(L, 512); else FRAD_GO(L, 1024); return 0;
    switch (c.log2m) {
        FRAD_CASE(6) FRAD_CASE(7) FRAD_CASE(8) FRAD_CASE(9) FRAD_CASE(10) FRAD_CASE(11) FRAD_CASE(12) FRAD_CASE(13)
        default: return -2;
    }
#undef FRAD_CASE
#undef FRAD_GO
}


}  // namespace frad
