// base-path kernels, K = 3, 16 steps per thread (windows of 3073..4096 steps), in a translation unit of their own (they are
// the slowest instantiations to compile).  Until round 3 this unit kept a 64-entry exp table while every other kernel had
// 256 entries (the backend's register-allocation fault of DESIGN.md section 5a struck its capped flavour with the larger
// one); since round 4 every kernel uses the same replicated 64-entry table (gibbs_device.hpp, exp_tab).
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant k3l16[] = { HMCG_V3(3, 16, false, false, H, P2) };
HMCG_GROUP(g_group_k3_l16, k3l16);
}
