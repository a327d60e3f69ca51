// D2H copy rate beside a running kernel (tools/ubench: diagnostics, not part of the library).
// build: hipcc --offload-arch=gfx950 -O2 -o d2h_rate d2h_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void __launch_bounds__(512) spin(long long ticks, double* sink)
{
    const long long t0 = __builtin_readcyclecounter();
    double a = threadIdx.x;
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[0] = a;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t MAXB = 64u << 20;
    char *d = nullptr, *p = nullptr; double* sink = nullptr;
    CK(hipMalloc(&d, MAXB)); CK(hipHostMalloc(&p, MAXB, hipHostMallocDefault)); CK(hipMalloc(&sink, 8));
    memset(p, 1, MAXB);
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const double mbs[] = {1.3, 2.6, 5.1, 10.2, 20.5, 41.0};
    for (int busy = 0; busy < 3; ++busy) {
        for (double mb : mbs) {
            const size_t bytes = (size_t)(mb * 1e6);
            double best = 1e9, sum = 0; const int reps = 6;
            for (int r = 0; r < reps; ++r) {
                // busy 1: 256 blocks x 512 threads for ~6 ms (100 MHz ticks of s_memtime... readcyclecounter = shader clock); busy 2: 1024 blocks
                if (busy) { hipLaunchKernelGGL(spin, dim3(busy == 1 ? 256 : 2048), dim3(512), 0, sk, 12000000LL, sink); }
                if (busy) { std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count() < 0.5) {} }
                const double t0 = now();
                CK(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, sc));
                CK(hipEventRecord(e, sc));
                CK(hipEventSynchronize(e));
                const double t1 = now();
                CK(hipStreamSynchronize(sk));
                if (r > 0) { best = t1 - t0 < best ? t1 - t0 : best; sum += t1 - t0; }
            }
            printf("busy=%d  %5.1f MB: mean %.3f ms (%.1f GB/s)  best %.3f ms (%.1f GB/s)\n", busy, mb, sum / (reps - 1), mb / (sum / (reps - 1)), best, mb / best);
        }
    }
    return 0;
}
