// variants.hpp -- the table of compiled kernel instantiations, split over several translation units
// (variants_*.hip) so that they compile in parallel; hmcg.hip only sees the tables.
#pragma once
#include "gibbs_device.hpp"

namespace hmcg_host {

using KernelFn = void (*)(const hmcg::KernelParams);

struct Variant {
    int K, L, NT;
    KernelFn fn;
    bool sig, smooth;
    int NH;                // helper waves on top of the NT window threads (block = NT + 64*NH threads)
    int occ;               // 2: registers capped so that two plain blocks share a CU
    int pref_small;        // flavour to run when every window has a CU to itself (W <= CU count)
    int pref_big;          // flavour for larger batches
};
// flavours: P1 = plain, whole register file; P2 = plain, two blocks per CU; H = four helper waves
enum { P1 = 0, P2 = 1, H = 2 };

struct VariantGroup {
    const Variant* v;
    int n;
};
// register-resident kernels (gibbs_device.hpp): base path by K, signal path, smoothed-probability path
extern const VariantGroup g_group_k2, g_group_k3, g_group_mid, g_group_k3_l16, g_group_k4, g_group_sig, g_group_smooth, g_group_sigsmooth;

using BigKernelFn = void (*)(const hmcg::KernelParams, const int);
struct BigVariant {
    int K, NT;
    BigKernelFn fn;
};
// LDS-resident kernels (gibbs_big.hpp): large K, or windows too long for the register-resident variants
extern const BigVariant g_big_variants[], g_big_smooth_variants[], g_big_stream_variants[], g_big_sig_variants[], g_big_sig_stream_variants[],
    g_big_sigsmooth_variants[], g_big_smooth_stream_variants[], g_big_sigsmooth_stream_variants[];
extern const int g_n_big_variants, g_n_big_smooth_variants, g_n_big_stream_variants, g_n_big_sig_variants, g_n_big_sig_stream_variants,
    g_n_big_sigsmooth_variants, g_n_big_smooth_stream_variants, g_n_big_sigsmooth_stream_variants;

#define HMCG_V(K_, L_, NT_, SIG_, SM_, NH_, OCC_, PS_, PB_) \
    { K_, L_, NT_, hmcg::gibbs_sweeps_kernel<K_, L_, NT_, SIG_, SM_, NH_, OCC_>, SIG_, SM_, NH_, OCC_, PS_, PB_ }
// every 256-thread variant in the three flavours, with the one to prefer for small and for large batches
// (measured: tools/variant_sweep.py, profiles/r01/variant_sweep.txt -- helper waves win while they fit the
// 256-register cap without spilling, capped plain blocks win once two windows can share a CU)
#define HMCG_V3(K_, L_, SIG_, SM_, PS_, PB_)                                                   \
    HMCG_V(K_, L_, 256, SIG_, SM_, 0, 1, PS_, PB_), HMCG_V(K_, L_, 256, SIG_, SM_, 0, 2, PS_, PB_), \
    HMCG_V(K_, L_, 256, SIG_, SM_, 4, 2, PS_, PB_)
#define HMCG_GROUP(name_, array_) const VariantGroup name_ = { array_, (int)(sizeof(array_) / sizeof(array_[0])) }

}  // namespace hmcg_host
