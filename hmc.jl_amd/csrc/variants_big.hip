// LDS-resident kernels: K = 5..8, and K = 2..4 for windows too long for the register-resident variants
#include <hip/hip_runtime.h>
#include "variants.hpp"
#include "gibbs_big.hpp"
namespace hmcg_host {
const BigVariant g_big_variants[] = {
    { 2, 256, hmcg::gibbs_sweeps_kernel_big<2, 256> }, { 3, 256, hmcg::gibbs_sweeps_kernel_big<3, 256> },
    { 4, 256, hmcg::gibbs_sweeps_kernel_big<4, 256> },
    { 5, 256, hmcg::gibbs_sweeps_kernel_big<5, 256> }, { 6, 256, hmcg::gibbs_sweeps_kernel_big<6, 256> },
    { 7, 256, hmcg::gibbs_sweeps_kernel_big<7, 256> }, { 8, 256, hmcg::gibbs_sweeps_kernel_big<8, 256> },
};
const int g_n_big_variants = (int)(sizeof(g_big_variants) / sizeof(g_big_variants[0]));
// ... with the backward pass every kept sweep (smoothed / filtered probability means streamed through HBM)
const BigVariant g_big_smooth_variants[] = {
    { 2, 256, hmcg::gibbs_sweeps_kernel_big<2, 256, true> }, { 3, 256, hmcg::gibbs_sweeps_kernel_big<3, 256, true> },
    { 4, 256, hmcg::gibbs_sweeps_kernel_big<4, 256, true> },
    { 5, 256, hmcg::gibbs_sweeps_kernel_big<5, 256, true> }, { 6, 256, hmcg::gibbs_sweeps_kernel_big<6, 256, true> },
    { 7, 256, hmcg::gibbs_sweeps_kernel_big<7, 256, true> }, { 8, 256, hmcg::gibbs_sweeps_kernel_big<8, 256, true> },
};
const int g_n_big_smooth_variants = (int)(sizeof(g_big_smooth_variants) / sizeof(g_big_smooth_variants[0]));
}
