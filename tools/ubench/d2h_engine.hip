// Which engine carries a D2H hipMemcpyAsync?  Run under `rocprofv3 --kernel-trace --memory-copy-trace`: an SDMA copy shows as a
// MEMORY_COPY row, a shader copy as a __amd_rocclr_copyBuffer kernel.  Each variant copies a different, recognisable size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fill(double* p, size_t n, double v) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main()
{
    const size_t MB = 1000000;
    char *d = nullptr, *p = nullptr;
    CK(hipMalloc(&d, 200 * MB)); CK(hipHostMalloc(&p, 200 * MB, hipHostMallocDefault));
    memset(p, 1, 200 * MB);
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t e, ek; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ek, hipEventDisableTiming));
    // V0 (11 MB): plain copy on an idle copy stream, source never touched by a kernel
    CK(hipMemcpyAsync(p, d, 11 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
    // V1 (12 MB): source written by a kernel on another stream, host waits for that kernel, then copies on the copy stream
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, sk, (double*)d, 12 * MB / 8, 1.0); CK(hipEventRecord(ek, sk)); CK(hipEventSynchronize(ek));
    CK(hipMemcpyAsync(p, d, 12 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
    // V2 (13 MB): as V1 but the copy stream waits for the kernel's event on the device
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, sk, (double*)d, 13 * MB / 8, 2.0); CK(hipEventRecord(ek, sk)); CK(hipStreamWaitEvent(sc, ek, 0));
    CK(hipMemcpyAsync(p, d, 13 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
    // V3 (14 MB): a small D2H first on the copy stream, then the big one (as the library's first chunk: status words, then the chunk)
    CK(hipMemcpyAsync(p + 150 * MB, d + 150 * MB, 1024, hipMemcpyDeviceToHost, sc));
    CK(hipMemcpyAsync(p, d, 14 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
    // V4 (15 MB): copy on the stream the kernel ran on, right behind it
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, sk, (double*)d, 15 * MB / 8, 3.0);
    CK(hipMemcpyAsync(p, d, 15 * MB, hipMemcpyDeviceToHost, sk)); CK(hipStreamSynchronize(sk));
    // V5 (16 MB): destination at an odd offset inside the pinned block, source at an odd offset inside the device block
    CK(hipMemcpyAsync(p + 20 * MB + 1048, d + 30 * MB + 2072, 16 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
    // V6 (17 MB): while a kernel is RUNNING on the other stream (enqueue a long fill first)
    hipLaunchKernelGGL(fill, dim3(64), dim3(64), 0, sk, (double*)(d + 100 * MB), 90 * MB / 8, 4.0);
    CK(hipMemcpyAsync(p, d, 17 * MB, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sk));
    printf("done\n");
    return 0;
}
