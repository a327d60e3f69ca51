// Issue cost of the fp64-side instructions the K = 8 kernel is made of, one wave per SIMD on every CU (the kernel's own
// occupancy): 32 independent instances of one instruction per round, straight-line, s_memtime ticks per instruction.
// Asked in round 4: the hand-scheduled chunk product runs its v_fma_f64 at ~5 ticks and the staged pdf pass at ~6.7 --
// which instructions are not full-rate?
//   hipcc --offload-arch=gfx950 -O3 -o op_issue tools/ubench/op_issue.hip && ./op_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define NI 32
template <int OP> __global__ __launch_bounds__(256) void k(unsigned long long* out, double* sink, const double* src, int e)
{
    double a[NI], x[NI];
    int n[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { a[i] = src[i] + threadIdx.x * 1e-3; x[i] = src[32 + i] * (1 + 1e-3 * threadIdx.x); n[i] = i + (int)threadIdx.x; }
    const double su = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(src[70])), __builtin_amdgcn_readfirstlane(__double2loint(src[70])));
    const int se = __builtin_amdgcn_readfirstlane(e);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (OP == 0) a[i] = fma(x[i], x[(i + 1) & 31], a[i]);                      // v_fma_f64, three VGPR operands
            if (OP == 1) a[i] = fma(x[i], su, a[i]);                                    // v_fma_f64, one SGPR operand
            if (OP == 2) a[i] = a[i] * x[i];                                            // v_mul_f64
            if (OP == 3) a[i] = a[i] + x[i];                                            // v_add_f64
            if (OP == 4) a[i] = fmax(a[i], x[i]);                                       // v_max_f64
            if (OP == 5) a[i] = ldexp(a[i], se);                                        // v_ldexp_f64 (scalar exponent)
            if (OP == 6) a[i] = ldexp(a[i], n[i]);                                      // v_ldexp_f64 (vector exponent)
            if (OP == 7) a[i] = rint(a[i] + x[i]);                                      // v_add + v_rndne_f64
            if (OP == 8) { n[i] += (int)a[i]; }                                         // v_cvt_i32_f64 + v_add_u32
            if (OP == 9) asm volatile("v_accvgpr_write_b32 a%1, %0\n" :: "v"(n[i]), "n"(i));   // v_accvgpr_write
            if (OP == 10) n[i] = max(n[i], max(n[(i + 1) & 31], n[(i + 2) & 31]));     // v_max3
            if (OP == 11) a[i] = __builtin_amdgcn_rcp(a[i]);                            // v_rcp_f64
            if (OP == 12) a[i] = fma(a[i], 0.5, 1.0);                                   // v_fma_f64 inline constants
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += a[i] + n[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int OP> void run(const char* name, double per = 1.0)
{
    unsigned long long* d; double *s, *src;
    hipMalloc(&d, 4096); hipMalloc(&s, 8 << 20); hipMalloc(&src, 1024);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, s, src, 1);
    hipDeviceSynchronize();
    unsigned long long t[4]; hipMemcpy(t, d, sizeof t, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f ticks per round-instruction (%s)\n", name, (double)t[0] / (REP * (double)NI), per > 1 ? "two instructions" : "one");
    hipFree(d); hipFree(s); hipFree(src);
}
int main()
{
    run<0>("v_fma_f64 v, v, v"); run<1>("v_fma_f64 v, s, v"); run<12>("v_fma_f64 v, const, const"); run<2>("v_mul_f64"); run<3>("v_add_f64");
    run<4>("v_max_f64"); run<5>("v_ldexp_f64 (scalar exp)"); run<6>("v_ldexp_f64 (vector exp)"); run<7>("v_add_f64 + v_rndne_f64", 2);
    run<8>("v_cvt_i32_f64 + v_add_u32", 2); run<9>("v_accvgpr_write_b32"); run<10>("v_max3_i32"); run<11>("v_rcp_f64");
    return 0;
}
