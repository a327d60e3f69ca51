// base-path kernels, K = 3 (the headline configuration runs <3, 4, 256> with helper waves)
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant k3[] = {
    HMCG_V3(3, 1, false, false, H, P2), HMCG_V3(3, 2, false, false, H, P1), HMCG_V3(3, 4, false, false, H, P2),
    HMCG_V3(3, 8, false, false, H, P2), HMCG_V3(3, 16, false, false, P1, P2),
    HMCG_V(3, 2, 512, false, false, 0, 1, P1, P1), HMCG_V(3, 8, 128, false, false, 0, 2, P2, P2),
};
HMCG_GROUP(g_group_k3, k3);
}
