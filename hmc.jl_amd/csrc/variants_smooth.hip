// with the smoothed-probability output (full backward pass every sweep)
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant smooth[] = {
    HMCG_V3(2, 1, false, true, P1, P1), HMCG_V3(2, 2, false, true, H, P1), HMCG_V3(2, 4, false, true, H, P2),
    HMCG_V3(3, 1, false, true, H, P2), HMCG_V3(3, 2, false, true, H, P2), HMCG_V3(3, 4, false, true, P1, P2),
    HMCG_V3(3, 8, false, true, P1, P1),
    HMCG_V3(4, 1, false, true, P1, P2), HMCG_V3(4, 2, false, true, P1, P2), HMCG_V3(4, 4, false, true, P1, P1),
};
HMCG_GROUP(g_group_smooth, smooth);
}
