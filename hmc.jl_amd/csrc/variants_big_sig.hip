// LDS-resident kernel on the signal Monte-Carlo path (estimatesignals!, src/Hmc.jl:868-914): K = 5..8, and K = 2..4 for windows
// too long for the register-resident SIG variants; LDS-resident and HBM-streaming forms
#include <hip/hip_runtime.h>
#include "variants.hpp"
#include "gibbs_big.hpp"
namespace hmcg_host {
#define HMCG_BS(K_, STREAM_) { K_, 256, hmcg::gibbs_sweeps_kernel_big<K_, 256, false, STREAM_, true> }
#define HMCG_BSS(K_, STREAM_) { K_, 256, hmcg::gibbs_sweeps_kernel_big<K_, 256, true, STREAM_, true> }
const BigVariant g_big_sig_variants[] = { HMCG_BS(2, false), HMCG_BS(3, false), HMCG_BS(4, false), HMCG_BS(5, false),
                                          HMCG_BS(6, false), HMCG_BS(7, false), HMCG_BS(8, false) };
const int g_n_big_sig_variants = (int)(sizeof(g_big_sig_variants) / sizeof(g_big_sig_variants[0]));
const BigVariant g_big_sig_stream_variants[] = { HMCG_BS(2, true), HMCG_BS(3, true), HMCG_BS(4, true), HMCG_BS(5, true),
                                                 HMCG_BS(6, true), HMCG_BS(7, true), HMCG_BS(8, true) };
const int g_n_big_sig_stream_variants = (int)(sizeof(g_big_sig_stream_variants) / sizeof(g_big_sig_stream_variants[0]));
// ... with the smoothing pass (K <= 4 within the register-resident range run variants_sigsmooth.hip instead)
const BigVariant g_big_sigsmooth_variants[] = { HMCG_BSS(2, false), HMCG_BSS(3, false), HMCG_BSS(4, false), HMCG_BSS(5, false),
                                                HMCG_BSS(6, false), HMCG_BSS(7, false), HMCG_BSS(8, false) };
const int g_n_big_sigsmooth_variants = (int)(sizeof(g_big_sigsmooth_variants) / sizeof(g_big_sigsmooth_variants[0]));
}
