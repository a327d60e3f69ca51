#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <omp.h>
static inline double dev_round5_from_n(double n){ double q0=n*1e-5; double r=fma(-q0,1e5,n); return fma(r,1e-5,q0);} 
int main(){ long bad=0; 
#pragma omp parallel for reduction(+:bad) schedule(static)
 for(int64_t i=0;i<((int64_t)1<<32);++i){ double n=(double)i; if(dev_round5_from_n(n)!=n/1e5) bad++; }
 printf("exhaustive n<2^32: bad=%ld\n",bad);
 uint64_t s=88172645463325252ull; long bad2=0; for(long k=0;k<200000000;++k){ s^=s<<13; s^=s>>7; s^=s<<17; double n=(double)(s>>11); if(dev_round5_from_n(n)!=n/1e5) bad2++; }
 printf("random n<2^53: bad=%ld\n",bad2); return 0; }
