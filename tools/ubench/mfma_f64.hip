// fp64 MFMA issue rate on gfx950 against the fp64 VALU FMA rate of tools/ubench/fma_issue.hip (same tick: s_memtime):
// v_mfma_f64_16x16x4_f64 (1024 FMAs per wave-instruction) and v_mfma_f64_4x4x4_4b_f64 (4 blocks x 64 = 256 FMAs), NACC
// independent accumulators, one wave per SIMD.  Answers whether the matrix pipe would retire the K=8 chunk products'
// FMAs faster than the vector pipe does.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64 tools/ubench/mfma_f64.hip && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 256
template <int NACC, int KIND> __global__ __launch_bounds__(256) void k(unsigned long long* out, double* sink, const double* src)
{
    const double a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
    d4 acc16[NACC];
    double acc4[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc16[i] = d4{0.0, 0.0, 0.0, 0.0}; acc4[i] = 0.0; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] + acc4[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int NACC, int KIND> void run(const char* name, int fmas)
{
    unsigned long long* d; double *s, *src;
    hipMalloc(&d, 256); hipMalloc(&s, 8192); hipMalloc(&src, 1024);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int q = 0; q < 2; ++q) hipLaunchKernelGGL((k<NACC, KIND>), dim3(1), dim3(256), 0, 0, d, s, src);
    hipDeviceSynchronize();
    unsigned long long t[4]; hipMemcpy(t, d, sizeof t, hipMemcpyDeviceToHost);
    const double per = (double)t[0] / (REP * (double)NACC);
    printf("%-26s accumulators=%2d : %7.2f ticks per instruction = %6.3f ticks per 64 FMAs (one VALU FMA instruction's worth)\n",
           name, NACC, per, per * 64.0 / fmas);
    hipFree(d); hipFree(s); hipFree(src);
}
int main()
{
    run<1, 0>("v_mfma_f64_16x16x4_f64", 1024); run<2, 0>("v_mfma_f64_16x16x4_f64", 1024); run<4, 0>("v_mfma_f64_16x16x4_f64", 1024);
    run<8, 0>("v_mfma_f64_16x16x4_f64", 1024);
    run<1, 1>("v_mfma_f64_4x4x4_4b_f64", 256); run<2, 1>("v_mfma_f64_4x4x4_4b_f64", 256); run<4, 1>("v_mfma_f64_4x4x4_4b_f64", 256);
    run<8, 1>("v_mfma_f64_4x4x4_4b_f64", 256); run<16, 1>("v_mfma_f64_4x4x4_4b_f64", 256);
    return 0;
}
