// base-path kernels, K = 4
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant k4[] = {
    HMCG_V3(4, 1, false, false, H, P2), HMCG_V3(4, 2, false, false, H, P2), HMCG_V3(4, 4, false, false, H, P2),
    HMCG_V3(4, 8, false, false, P1, P2),
};
HMCG_GROUP(g_group_k4, k4);
}
