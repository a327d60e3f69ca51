// HBM-streaming forms of the LDS-resident kernel WITH the smoothing pass: plain and signal path
#include <hip/hip_runtime.h>
#include "variants.hpp"
#include "gibbs_big.hpp"
namespace hmcg_host {
#define HMCG_BX(K_, SIG_) { K_, 256, hmcg::gibbs_sweeps_kernel_big<K_, 256, true, true, SIG_> }
const BigVariant g_big_smooth_stream_variants[] = { HMCG_BX(2, false), HMCG_BX(3, false), HMCG_BX(4, false), HMCG_BX(5, false),
                                                    HMCG_BX(6, false), HMCG_BX(7, false), HMCG_BX(8, false) };
const int g_n_big_smooth_stream_variants = (int)(sizeof(g_big_smooth_stream_variants) / sizeof(g_big_smooth_stream_variants[0]));
const BigVariant g_big_sigsmooth_stream_variants[] = { HMCG_BX(2, true), HMCG_BX(3, true), HMCG_BX(4, true), HMCG_BX(5, true),
                                                       HMCG_BX(6, true), HMCG_BX(7, true), HMCG_BX(8, true) };
const int g_n_big_sigsmooth_stream_variants = (int)(sizeof(g_big_sigsmooth_stream_variants) / sizeof(g_big_sigsmooth_stream_variants[0]));
}
