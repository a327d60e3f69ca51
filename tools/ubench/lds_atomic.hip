// LDS atomic-add cost under same-address conflicts (64 lanes -> NADDR addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(unsigned long long* out, double* sink, int naddr)
{
    __shared__ double acc[4][64];
    __shared__ int cnt[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    acc[wave][lane] = 0; cnt[wave][lane] = 0;
    __syncthreads();
    const int a = (lane * 7 + lane / 5) % naddr;
    double v = 1.0 + lane * 1e-3;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
    for (int r = 0; r < 64; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) __hip_atomic_fetch_add(&acc[wave][a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(&cnt[wave][a], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __syncthreads();
    sink[threadIdx.x] = acc[wave][lane] + cnt[wave][lane];
    if (lane == 0) out[wave] = t1 - t0;
}
int main()
{
    unsigned long long* d; double* s; hipMalloc(&d, 64); hipMalloc(&s, 256 * 8);
    for (int mode = 0; mode < 2; ++mode)
        for (int naddr : {1, 3, 9, 64}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, s, naddr);
                else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, s, naddr);
                hipDeviceSynchronize();
            }
            unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            double hs[256]; hipMemcpy(hs, s, 2048, hipMemcpyDeviceToHost);
            printf("%s  %2d addresses: %7.1f cycles per wave-instruction (4 waves issuing concurrently; wave0 %llu wave3 %llu) check %.3f\n",
                   mode == 0 ? "ds_add_f64" : "ds_add_u32", naddr, (double)h[3] / 256.0, h[0], h[3], hs[0]);
        }
    return 0;
}
