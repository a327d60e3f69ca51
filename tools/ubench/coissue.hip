// Co-issue microbenchmark for gfx950: does a second wave on the same SIMD fill the issue slots a first wave leaves
// (a lone wave issues one fp64 op per ~9.6 cycles, the SIMD can take one per ~4.5) when each wave holds a LARGE
// register footprint and runs a long straight-line mix, as the sweep kernel's waves do?  Variants:
//   NV      live fp64 values per lane (register footprint ~ 2*NV VGPRs)
//   MIX 0   fma only; 1 fma + cndmask + int; 2 the exp_tab sequence (LDS table lookup); 3 fma + DPP mov
// Reports cycles per wave-instruction-group for 1 and 2 waves per SIMD (256 / 512 threads, one block).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int NV, int MIX> __global__ __launch_bounds__(512) void k(unsigned long long* out, double* sink, double a0, int sel)
{
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = 1.0 + threadIdx.x * 0.01;
    __syncthreads();
    double a[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = a0 + i * 0.37 + threadIdx.x * 1e-3;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (MIX == 0) a[i] = fma(a[i], 1.0000001, 0.5);
            if (MIX == 1) { a[i] = fma(a[i], 1.0000001, a[(i + 7) % NV]); a[i] = (a[(i + 3) % NV] > 2.0 + sel) ? a[i] : a[(i + 1) % NV]; }
            if (MIX == 2) {
                double x = -fabs(a[i]) * 1e-3;
                const double n = rint(x * 92.332482616893657);
                double rr = fma(-n, 1.0830424693267560e-02, x);
                rr = fma(-n, 2.9815858269852933e-12, rr);
                const int ni = (int)n;
                const double tj = tab[ni & 63];
                double p = fma(rr, 1.0 / 120.0, 1.0 / 24.0);
                p = fma(p, rr, 1.0 / 6.0); p = fma(p, rr, 0.5); p = fma(p, rr, 1.0); p = fma(p, rr, 1.0);
                a[i] = ldexp(tj * p, ni >> 6) + a[i];
            }
            if (MIX == 3) {
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a[i]), 0x111, 0xF, 0xF, false);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a[i]), 0x111, 0xF, 0xF, false);
                a[i] = fma(__hiloint2double(hi, lo), 0.5, a[(i + 5) % NV]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += a[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int NV, int MIX> void run(const char* name)
{
    unsigned long long* d; double* s; hipMalloc(&d, 256); hipMalloc(&s, 1024 * 8);
    hipFuncAttributes fa{}; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<NV, MIX>));
    printf("%-34s NV=%3d regs=%3d", name, NV, fa.numRegs);
    double per1 = 0;
    for (int nt = 256; nt <= 512; nt *= 2) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<NV, MIX>), dim3(1), dim3(nt), 0, 0, d, s, 1.5, 0); hipDeviceSynchronize(); }
        unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mn = ~0ull; for (int i = 0; i < nt / 64; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
        const double per = (double)mx / (REP * (double)NV);
        if (nt == 256) per1 = per;
        printf("  | %dw/SIMD: slowest %7.2f fastest %7.2f cyc/group", nt / 256, per, (double)mn / (REP * (double)NV));
        if (nt == 512) printf("  -> SIMD throughput x%.2f", 2.0 * per1 / per);
    }
    printf("\n");
    hipFree(d); hipFree(s);
}
int main()
{
    run<8, 0>("fma only"); run<100, 0>("fma only"); run<120, 0>("fma only");
    run<8, 1>("fma+cmp+cndmask"); run<100, 1>("fma+cmp+cndmask");
    run<8, 2>("exp_tab (LDS table)"); run<60, 2>("exp_tab (LDS table)"); run<100, 2>("exp_tab (LDS table)");
    run<8, 3>("2 dpp mov + fma"); run<100, 3>("2 dpp mov + fma");
    return 0;
}
