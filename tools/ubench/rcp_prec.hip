// Precision of the hardware v_rcp_f64 / v_rsq_f64 on gfx950 against correctly rounded 1/x and 1/sqrt(x) (long double on
// the host): max error in ulps over a log-uniform sample.  Decides whether the Newton steps in rcp_fast / rsqrt_fast
// are needed.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r, double* q, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), r(n), q(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 80.0) * (1.0 + u);
    }
    double *dx, *dr, *dq;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dr, n * 8); (void)hipMalloc(&dq, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dq, n);
    (void)hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
    double mr = 0, mq = 0;
    for (int i = 0; i < n; ++i) {
        const long double er = 1.0L / (long double)x[i], eq = 1.0L / sqrtl((long double)x[i]);
        const double ur = std::fabs((double)(((long double)r[i] - er) / er)) / 1.1102230246251565e-16;
        const double uq = std::fabs((double)(((long double)q[i] - eq) / eq)) / 1.1102230246251565e-16;
        if (ur > mr) mr = ur;
        if (uq > mq) mq = uq;
    }
    printf("v_rcp_f64 max rel err = %.3g x 2^-53   v_rsq_f64 max rel err = %.3g x 2^-53\n", mr, mq);
    return 0;
}
