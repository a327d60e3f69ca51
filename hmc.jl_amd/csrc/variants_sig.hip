// signal Monte-Carlo path (estimatesignals!): two-population statistics, per-step emission scale
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant sig[] = {
    HMCG_V3(2, 1, true, false, H, P1), HMCG_V3(2, 2, true, false, H, P1), HMCG_V3(2, 4, true, false, H, P1),
    HMCG_V3(3, 1, true, false, H, P1), HMCG_V3(3, 2, true, false, H, P2), HMCG_V3(3, 4, true, false, H, P2),
    HMCG_V3(3, 8, true, false, H, P2),
    HMCG_V3(4, 1, true, false, H, P2), HMCG_V3(4, 2, true, false, H, P2), HMCG_V3(4, 4, true, false, H, P2),
};
HMCG_GROUP(g_group_sig, sig);
}
