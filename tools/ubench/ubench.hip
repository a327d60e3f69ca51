// Instruction-issue microbenchmark for gfx950: one wave per SIMD (256-thread block, 1 block),
// N independent chains of each op in an unrolled loop; reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP 256
template <int OP> __global__ __launch_bounds__(1024) void k(unsigned long long* out, double* sink, double a0, unsigned u0)
{
    double a[8]; unsigned u[8]; unsigned long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i + threadIdx.x * 1e-3; u[i] = u0 + i * 77u + threadIdx.x; w[i] = u[i]; }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = fma(a[i], 1.0000001, 0.5);
            if (OP == 1) a[i] = a[i] * 1.0000001;
            if (OP == 2) a[i] = a[i] + 0.25;
            if (OP == 3) w[i] = (unsigned long long)(unsigned)w[i] * 0xD2511F53u + (w[i] >> 32);
            if (OP == 4) u[i] = __umulhi(u[i], 0xD2511F53u) ^ 0x9E3779B9u;
            if (OP == 5) u[i] = u[i] * 0xCD9E8D57u + 1u;
            if (OP == 6) u[i] = __builtin_amdgcn_alignbit(u[i], u[i], 13) + 0x9E3779B9u;
            if (OP == 7) u[i] = (u[i] ^ (u[i] >> 7)) + 3u;
            if (OP == 8) a[i] = fmax(a[i], 0.75 * i);
            if (OP == 9) a[i] = ldexp(a[i], 1);
            if (OP == 10) u[i] = __builtin_amdgcn_update_dpp(0, (int)u[i], 0x111, 0xF, 0xF, false) + 1;
            if (OP == 11) u[i] = __shfl_up(u[i], 1, 64) + 1;
            if (OP == 12) a[i] = (u[i] & 1) ? a[i] : a[(i + 1) & 7] + 1.0;
            if (OP == 13) u[i] = __builtin_amdgcn_readlane((int)u[i], 5) + threadIdx.x;
            if (OP == 14) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.0;
            if (OP == 15) a[i] = sqrt(a[i]) + 1.0;
            if (OP == 16) a[i] = 1.0 / a[i] + 1.0;
            if (OP == 17) u[i] = __mul24((int)(u[i] & 0xFFFFFF), 0x51F53) + 1u;
            if (OP == 18) u[i] = __popcll(__ballot(u[i] & 1)) + u[i];
            if (OP == 19) a[i] = (double)u[i] + a[i];
            if (OP == 20) a[i] = rint(a[i]) * 0.999 + 0.1;
            if (OP == 21) u[i] = __builtin_amdgcn_update_dpp(0, (int)u[i], 0x142, 0xA, 0xF, false) + 1;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + u[i] + (double)w[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int OP> void run(const char* name)
{
    unsigned long long* d; double* s; hipMalloc(&d, 256); hipMalloc(&s, 1024 * 8);
    printf("%-28s", name);
    for (int nt = 256; nt <= 1024; nt *= 2) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(nt), 0, 0, d, s, 1.5, 12345u); hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(nt), 0, 0, d, s, 1.5, 12345u); hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < nt / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("  %dw/SIMD: %6.2f (%5.2f/wave)", nt / 256, (double)mx / (REP * 8.0), (double)mx / (REP * 8.0) / (nt / 256));
    }
    printf("   (slowest wave; per-wave-instruction throughput in brackets)\n");
    hipFree(d); hipFree(s);
}
int main()
{
    run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("mad_u64_u32 (mul 32x32->64)");
    run<4>("v_mul_hi_u32 + xor"); run<5>("v_mul_lo_u32 + add"); run<6>("alignbit + add"); run<7>("xor,shift,add (3 int ops)");
    run<8>("v_max_f64"); run<9>("v_ldexp_f64"); run<10>("dpp row_shr mov + add"); run<11>("shfl_up(bpermute) + add");
    run<12>("cndmask f64 + add_f64"); run<13>("readlane + add"); run<14>("v_rcp_f64 + add"); run<15>("sqrt(f64) + add");
    run<16>("1.0/x (IEEE div) + add"); run<17>("mul_u24 + and + add"); run<18>("ballot+bcnt+add"); run<19>("cvt_f64_u32 + add_f64");
    run<20>("rint + mul + add (f64)"); run<21>("dpp row_bcast15 mov + add");
    return 0;
}
