// checks that device fp32 / fp64 primitives are bit-identical to the host's (IEEE, no contraction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include "../include/suma_detmath.h"
#define N (1<<20)
__global__ void k(const float* a, const float* b, const float* c, const float* d, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= N) return;
  out[0*N+i] = a[i] / b[i];
  out[1*N+i] = sdm_sqrt(a[i] < 0 ? -a[i] : a[i]);
  out[2*N+i] = a[i] * b[i] - c[i] * d[i];
  out[3*N+i] = (a[i] * b[i] + c[i] * d[i]) + a[i] * d[i];
  out[4*N+i] = sdm_atan2(a[i], b[i]);
  out[5*N+i] = sdm_asin(a[i] / (sdm_abs(a[i]) + sdm_abs(b[i]) + 1e-3f));
  out[6*N+i] = sdm_exp(a[i]);
  out[7*N+i] = sdm_log(sdm_abs(b[i]) + 1e-6f);
  out[8*N+i] = (float)((double)a[i] * 268435456.0 + 6755399441055744.0 - 6755399441055744.0);
  out[9*N+i] = (float)(long long)(a[i]*1000.f);
  out[10*N+i] = (float)(sdm_sin_d((double)a[i]) + sdm_cos_d((double)b[i]));
  out[11*N+i] = (float)((double)a[i] / (double)b[i]) + (float)sdm_sqrt_d((double)sdm_abs(a[i]));
}
int main() {
  float *h[4], *dv[4], *dout, *hout = (float*)malloc(12*N*4);
  srand(1);
  for (int j = 0; j < 4; ++j) { h[j] = (float*)malloc(N*4); for (int i = 0; i < N; ++i) h[j][i] = ((rand() / (float)RAND_MAX) * 2 - 1) * ((i & 7) == 0 ? 50.f : 1.f);
    hipMalloc(&dv[j], N*4); hipMemcpy(dv[j], h[j], N*4, hipMemcpyHostToDevice); }
  hipMalloc(&dout, 12*N*4);
  k<<<N/256, 256>>>(dv[0], dv[1], dv[2], dv[3], dout);
  hipMemcpy(hout, dout, 12*N*4, hipMemcpyDeviceToHost);
  const char* names[12] = {"div","sqrt","mulsub","dot","atan2","asin","exp","log","magic","f2ll","sincos_d","div_d"};
  for (int op = 0; op < 12; ++op) { long bad = 0; for (int i = 0; i < N; ++i) { float a=h[0][i],b=h[1][i],c=h[2][i],d=h[3][i], r;
      switch(op){case 0: r=a/b;break; case 1: r=sdm_sqrt(a<0?-a:a);break; case 2: r=a*b-c*d;break; case 3: r=(a*b+c*d)+a*d;break; case 4: r=sdm_atan2(a,b);break;
      case 5: r=sdm_asin(a/(sdm_abs(a)+sdm_abs(b)+1e-3f));break; case 6: r=sdm_exp(a);break; case 7: r=sdm_log(sdm_abs(b)+1e-6f);break;
      case 8: r=(float)((double)a*268435456.0+6755399441055744.0-6755399441055744.0);break; case 9: r=(float)(long long)(a*1000.f);break;
      case 10: r=(float)(sdm_sin_d((double)a)+sdm_cos_d((double)b));break; default: r=(float)((double)a/(double)b) + (float)sdm_sqrt_d((double)sdm_abs(a));}
      if (memcmp(&r, &hout[op*N+i], 4)) { if (bad < 2) printf("  %s a=%a b=%a host=%a dev=%a\n", names[op], a, b, r, hout[op*N+i]); ++bad; } }
    printf("%-8s mismatches: %ld\n", names[op], bad); }
  return 0;
}
