// In-kernel shader clock under a dense fp64 FMA load, and the fp64 FMA rate that goes with it.
//   clock = d(s_memtime) / d(s_memrealtime) * 100 MHz   (MI355X_MICROARCH.md, "DVFS give-back" item 6)
// Every CU runs one wave per SIMD (or two: second argument) of 64 independent FMA chains on non-trivial operands; the kernel
// is launched back to back for `seconds` and the clock of the launch in flight is printed as the run goes on, so that a
// ramp (or a give-back) shows.  Also printed: ticks per wave-FMA and the chip's FMA rate against the wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o clock tools/ubench/clock.hip && ./clock [seconds=4] [waves_per_simd=1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define NC 64
__global__ __launch_bounds__(512) void k(unsigned long long* out, double* sink, const double* src, int reps)
{
    double acc[NC], x[8], y[8];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = src[i] + threadIdx.x * 1e-3;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = src[64 + i] * (1.0 + 1e-9 * threadIdx.x); y[i] = src[80 + i]; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[i] = fma(x[(i + kk) & 7], y[kk], acc[i] * 0.999);   // 2 fp64 ops per entry
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) s += acc[i];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t o = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2;
        out[o] = t1 - t0; out[o + 1] = r1 - r0;
    }
}
int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 1;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount, threads = 256 * wps, reps = 3000;
    const size_t nw = (size_t)blocks * (threads / 64);
    unsigned long long* d; double *s, *src;
    hipMalloc(&d, nw * 16); hipMalloc(&s, (size_t)blocks * threads * 8); hipMalloc(&src, 1024);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.37 + i * 1.3e-3;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> t(nw * 2);
    const auto w0 = std::chrono::steady_clock::now();
    double next_print = 0.0;
    printf("%d CUs, %d wave(s) per SIMD, %d reps x %d fp64 ops per wave and launch (clockRate attr %.0f MHz)\n", blocks, wps, reps, 8 * NC * 2, pr.clockRate / 1e3);
    for (int it = 0;; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, s, src, reps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (el >= next_print || el >= seconds) {
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(t.data(), d, nw * 16, hipMemcpyDeviceToHost);
            std::vector<double> clk(nw), tk(nw);
            for (size_t i = 0; i < nw; ++i) { clk[i] = (double)t[2 * i] / (double)t[2 * i + 1] * 100.0; tk[i] = (double)t[2 * i]; }
            std::sort(clk.begin(), clk.end()); std::sort(tk.begin(), tk.end());
            const double ops = (double)reps * 8 * NC * 2;        // fp64 wave-instructions per wave (mul + fma)
            printf("t=%5.2fs launch %4d: kernel %.3f ms | clock MHz min %.0f median %.0f max %.0f | %.2f ticks per fp64 wave-op | %.1f TFLOP/s (fma = 2, mul = 1)\n",
                   el, it, ms, clk.front(), clk[nw / 2], clk.back(), tk[nw / 2] / ops,
                   (double)nw * reps * 8 * NC * 3 * 64 / (ms * 1e-3) / 1e12);
            fflush(stdout);
            next_print = el < 0.5 ? el + 0.1 : el + 0.5;
        }
        if (el >= seconds) break;
    }
    return 0;
}
