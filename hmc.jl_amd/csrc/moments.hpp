// moments.hpp -- host-side launchers of the draw-moments kernels (moments.hip): correlations between the per-draw outputs
// of a window (calccorr, src/Hmc.jl:1094-1163), accumulated over the chunks of a run.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace hmcg_host {

struct MomentsArgs {
    const double* mu; const double* sig2; const double* pi_end; const double* A; const double* fcast;   // device draw arrays
    double* mom;               // device, [W][moments_stride(K)]: carried from chunk to chunk
    long long nd, nd_ld;       // draws in this block and the arrays' leading dimension
    int W, K, H;
    bool first;                // first block of the run
};
int corr_columns(int K);                   // 3K + K^2 + 1: mu | sigma | pi | vec(A) | forecast of the first horizon
size_t moments_stride(int K);              // doubles per window in `mom`
hipError_t launch_moments(const MomentsArgs& a, hipStream_t stream);
hipError_t launch_corr_finalize(const double* mom, double* corr, int W, int K, hipStream_t stream);

}  // namespace hmcg_host
