// The steps-per-thread classes between the powers of two: a window of 513..768 steps runs three steps per thread instead of
// four (1025..1536: six instead of eight, 2049..3072: twelve instead of sixteen, K = 3).  The length-bucketed dispatch of hmcg.hip gives every window of a ragged
// batch the smallest class that covers it -- the reference's production windows of 513..579 months (code/run_hmm.jl:79-109)
// are the critical path of its 460-window run, on the base path and on the signal path alike.
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant mid[] = {
    HMCG_V3(3, 3, false, false, H, P2), HMCG_V3(3, 6, false, false, H, P2), HMCG_V3(3, 12, false, false, H, P2),
    HMCG_V3(2, 3, false, false, H, P1), HMCG_V3(4, 3, false, false, H, P2),
    HMCG_V3(3, 3, true, false, H, P2),
};
HMCG_GROUP(g_group_mid, mid);
}
