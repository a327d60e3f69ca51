// base-path kernels, K = 3, 16 steps per thread (windows of 2049..4096 steps), in a translation unit of their own: with the
// 256-entry exp table the capped flavour <3,16,256,...,0,2> comes out of the backend with register spills placed ahead of a
// join block's exec restore (the fault of DESIGN.md section 5a; tools/isa_lint.py refuses it, `make repro-exptab256` shows
// it), with the 64-entry table it does not -- so these three keep the smaller table.
#define HMCG_EXPTAB_N 64
#include <hip/hip_runtime.h>
#include "variants.hpp"
namespace hmcg_host {
static const Variant k3l16[] = { HMCG_V3(3, 16, false, false, P1, P2) };
HMCG_GROUP(g_group_k3_l16, k3l16);
}
