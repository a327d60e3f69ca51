// round5.hpp -- the reference's 5-digit rounding of every stored value, shared by the sweep kernels (on-device summary
// means) and the draw-moments kernel (correlations of the rounded draws).
#pragma once
#include <hip/hip_runtime.h>

namespace hmcg {

// Julia round(x; digits=5) (basicsave, src/Hmc.jl:719) = rint(x * 1e5) / 1e5 with a CORRECTLY ROUNDED quotient, as the
// oracle and a host-side mean of the per-draw CSV cells compute it: reciprocal estimate q0 = fl(n * 1e-5), exact fp64
// residual r = n - q0 * 1e5 (one FMA), correction q0 + r * 1e-5.  For |n| < 2^53 the corrected value is the correctly
// rounded n / 1e5 (Markstein's division sequence with an exact residual; 1e-5 is 1/1e5 to half an ulp).
__device__ __forceinline__ double round5(double x)
{
    const double n = rint(x * 1e5);
    const double q0 = n * 1e-5;
    const double r = fma(-q0, 1e5, n);
    const double q = fma(r, 1e-5, q0);
    return isfinite(q) ? q : x;
}

}  // namespace hmcg
