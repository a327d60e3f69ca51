// fp64 FMA issue rate of a lone wave on gfx950 as a function of (a) the number of independent accumulation chains and
// (b) where the multiplicands live: three VGPR operands (acc += x[i] * y[j], what the chunk products do) against one VGPR
// operand and inline constants.  Straight-line code (the 64-step outer loop is the only branch), s_memtime ticks per FMA.
//   hipcc --offload-arch=gfx950 -O3 -o fma_issue tools/ubench/fma_issue.hip && ./fma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 32
template <int NC, int MODE> __global__ __launch_bounds__(256) void k(unsigned long long* out, double* sink, const double* src)
{
    double acc[NC], x[8], y[8];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = src[i] + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = src[64 + i] * (1 + threadIdx.x); y[i] = src[80 + i]; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (MODE == 0) acc[i] = fma(x[(i + k) & 7], y[k], acc[i]);        // three VGPR operands
                if (MODE == 1) acc[i] = fma(acc[i], 0.5, 1.0);                    // one VGPR operand, inline constants
                if (MODE == 2) acc[i] = fma(x[(i + k) & 7], 0.5, acc[i]);         // two VGPR operands
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) s += acc[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NC, int MODE> void run(const char* name, int threads)
{
    unsigned long long* d; double *s, *src;
    hipMalloc(&d, 4096); hipMalloc(&s, 1 << 20); hipMalloc(&src, 1024);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<NC, MODE>), dim3(1), dim3(threads), 0, 0, d, s, src);
    hipLaunchKernelGGL((k<NC, MODE>), dim3(1), dim3(threads), 0, 0, d, s, src);
    hipDeviceSynchronize();
    unsigned long long t[4]; hipMemcpy(t, d, sizeof t, hipMemcpyDeviceToHost);
    printf("%-28s chains=%2d waves/SIMD=%d : %6.2f ticks per FMA\n", name, NC, threads > 256 ? 2 : 1, (double)t[0] / (REP * 8.0 * NC));
    hipFree(d); hipFree(s); hipFree(src);
}
int main()
{
    run<2, 0>("3 VGPR operands", 256); run<4, 0>("3 VGPR operands", 256); run<8, 0>("3 VGPR operands", 256);
    run<16, 0>("3 VGPR operands", 256); run<32, 0>("3 VGPR operands", 256); run<64, 0>("3 VGPR operands", 256);
    run<2, 1>("1 VGPR + constants", 256); run<4, 1>("1 VGPR + constants", 256); run<8, 1>("1 VGPR + constants", 256);
    run<16, 1>("1 VGPR + constants", 256); run<32, 1>("1 VGPR + constants", 256); run<64, 1>("1 VGPR + constants", 256);
    run<8, 2>("2 VGPR operands", 256); run<16, 2>("2 VGPR operands", 256); run<32, 2>("2 VGPR operands", 256);

    return 0;
}
