// Does the fp64 matrix pipe of gfx950 run BESIDE the fp64 vector pipe?  ("the matrix pipe as a second
// pipe".)  Same tick as fma_issue.hip / mfma_f64.hip (s_memtime).
//  A. one wave per SIMD, each loop round = NM MFMAs (independent accumulators) + NF independent v_fma_f64: if the pipes
//     overlap, ticks per round ~ max(NM * t_mfma, NF * 4.1); if they share the fp64 datapath, ~ the sum.
//  B. two waves per SIMD (512 threads): waves 0-3 issue only MFMAs, waves 4-7 only FMAs; each side's ticks against its
//     lone-wave figure.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue tools/ubench/mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 128
// KIND 0: v_mfma_f64_16x16x4_f64 (1024 FMAs), 1: v_mfma_f64_4x4x4_4b_f64 (256 FMAs)
template <int KIND, int NM, int NF, int SPLIT> __global__ __launch_bounds__(512) void k(unsigned long long* out, double* sink, const double* src)
{
    const double a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
    constexpr int NMA = NM > 0 ? NM : 1, NFA = NF > 0 ? NF : 1;
    d4 acc16[NMA];
    double acc4[NMA], f[NFA], x[8];
#pragma unroll
    for (int i = 0; i < NMA; ++i) { acc16[i] = d4{0.0, 0.0, 0.0, 0.0}; acc4[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NFA; ++i) f[i] = src[i] + threadIdx.x * 1e-3;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = src[96 + i] * (1.0 + 1e-9 * threadIdx.x);
    const bool do_m = SPLIT ? (threadIdx.x < 256) : true, do_f = SPLIT ? (threadIdx.x >= 256) : true;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (SPLIT) {
        if (do_m) {
#pragma unroll 1
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    if (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
                    else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
                }
        }
        if (do_f) {
#pragma unroll 1
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int i = 0; i < NF; ++i) f[i] = fma(f[i], x[i & 7], x[(i + 3) & 7]);
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < REP; ++r) {
            // interleave: the FMAs spread evenly between the MFMAs
#pragma unroll
            for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
                if (NM > 0) {
                    if (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
                    else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
                }
                constexpr int per = NF / (NM > 0 ? NM : 1);
#pragma unroll
                for (int j = 0; j < per; ++j) { const int q = i * per + j; f[q] = fma(f[q], x[q & 7], x[(q + 3) & 7]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NMA; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] + acc4[i];
#pragma unroll
    for (int i = 0; i < NFA; ++i) s += f[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND, int NM, int NF, int SPLIT> void run()
{
    unsigned long long* d; double *s, *src;
    hipMalloc(&d, 256); hipMalloc(&s, 8192); hipMalloc(&src, 1024);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    const int nt = SPLIT ? 512 : 256;
    for (int q = 0; q < 2; ++q) hipLaunchKernelGGL((k<KIND, NM, NF, SPLIT>), dim3(1), dim3(nt), 0, 0, d, s, src);
    hipDeviceSynchronize();
    unsigned long long t[8]; hipMemcpy(t, d, sizeof t, hipMemcpyDeviceToHost);
    const char* nm = KIND == 0 ? "16x16x4" : "4x4x4_4b";
    const int fm = KIND == 0 ? 1024 : 256;
    if (SPLIT)
        printf("two waves/SIMD  %-8s: MFMA wave %7.1f ticks/round (%d MFMA)   FMA wave %7.1f ticks/round (%d FMA = %.2f each)\n", nm,
               (double)t[0] / REP, NM, (double)t[4] / REP, NF, (double)t[4] / REP / (NF > 0 ? NF : 1));
    else
        printf("one wave        %-8s: %2d MFMA + %3d FMA per round: %7.1f ticks/round  -> %6.3f ticks per 64 FMAs overall\n", nm, NM, NF,
               (double)t[0] / REP, (double)t[0] / REP * 64.0 / ((double)NM * fm + 64.0 * NF));
    hipFree(d); hipFree(s); hipFree(src);
}
int main()
{
    run<0, 0, 64, 0>(); run<0, 4, 0, 0>();
    run<0, 4, 32, 0>(); run<0, 4, 64, 0>(); run<0, 4, 96, 0>(); run<0, 4, 128, 0>();
    run<1, 16, 0, 0>(); run<1, 16, 16, 0>(); run<1, 16, 32, 0>(); run<1, 16, 64, 0>();
    run<0, 4, 128, 1>(); run<1, 16, 64, 1>();
    return 0;
}
