// LDS-resident kernel's HBM-streaming form: the per-step arrays in a global scratch -- windows beyond what the CU's LDS holds
#include <hip/hip_runtime.h>
#include "variants.hpp"
#include "gibbs_big.hpp"
namespace hmcg_host {
const BigVariant g_big_stream_variants[] = {
    { 2, 256, hmcg::gibbs_sweeps_kernel_big<2, 256, false, true> }, { 3, 256, hmcg::gibbs_sweeps_kernel_big<3, 256, false, true> },
    { 4, 256, hmcg::gibbs_sweeps_kernel_big<4, 256, false, true> }, { 5, 256, hmcg::gibbs_sweeps_kernel_big<5, 256, false, true> },
    { 6, 256, hmcg::gibbs_sweeps_kernel_big<6, 256, false, true> }, { 7, 256, hmcg::gibbs_sweeps_kernel_big<7, 256, false, true> },
    { 8, 256, hmcg::gibbs_sweeps_kernel_big<8, 256, false, true> },
};
const int g_n_big_stream_variants = (int)(sizeof(g_big_stream_variants) / sizeof(g_big_stream_variants[0]));
}
