// base-path kernels, K = 2
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant k2[] = {
    HMCG_V3(2, 1, false, false, H, P2), HMCG_V3(2, 2, false, false, H, P1), HMCG_V3(2, 4, false, false, H, P1),
    HMCG_V3(2, 8, false, false, H, P2),
};
HMCG_GROUP(g_group_k2, k2);
}
