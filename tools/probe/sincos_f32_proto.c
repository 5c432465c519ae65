// DEV TOOL (round 6): CPU emulation of the packed-f32 sin / cos of csrc/elementwise.hip (sincos_f32_pair) — the SAME operation sequence, fmaf = one
// rounding — against f64 libm rounded once (the oracle's definition), for EVERY f32 with |x| < bound, sin and cos: histogram of ULP distances,
// largest error against the true value, zero-sign mismatches.  Coefficients: tools/probe/sincos_f32_fit.py 0.89.
//   gcc -O2 -march=native -ffp-contract=off -fopenmp tools/probe/sincos_f32_proto.c -o /tmp/proto -lm && /tmp/proto 1000000   (≈ 30 s on 8 cores)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <omp.h>
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
#ifndef S1
#define S1 -0x1.555534p-3f
#define S2 0x1.110122p-7f
#define S3 -0x1.976586p-13f
#define C1 0x1.55553ep-5f
#define C2 -0x1.6c0536p-10f
#define C3 0x1.982456p-16f
#endif
static inline float sincos_f32(float x, int want_cos){
  const float M = 12582912.0f;
  const float s = fmaf(x, 0x1.45f306p-1f, M);
  const float kf = s - M;
  const uint32_t q = f2u(s) + (uint32_t)want_cos;
  const float P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f, P3 = -0x1.ee59dap-50f;
  const float r1 = fmaf(kf, -P1, x);
  const float ph = fmaf(kf, P2, 0.0f);
  const float pl = fmaf(kf, P2, -ph);
  const float rh = r1 - ph;
  const float t = rh - r1;
  const float e = -ph - t;
  float nlo = pl - e;
  nlo = fmaf(kf, P3, nlo);
  const float z = rh * rh;
  float ps = fmaf(z, -(S3), -(S2));
  ps = fmaf(z, ps, -(S1));   // = -ps(z) > 0
  const float ws = rh * z;
  const float u = fmaf(ws, ps, nlo);
  const float S = rh - u;
  float pc = fmaf(z, C3, C2);
  pc = fmaf(z, pc, C1);
  pc = fmaf(z, pc, -0.5f);
  const float rl = rh * nlo;   // = -rh*lo
  const float tc = fmaf(z, pc, rl);
  const float C = 1.0f + tc;
  float res = (q & 1) ? C : S;
  return u2f(f2u(res) ^ ((q << 30) & 0x80000000u));
}
int main(int argc,char**argv){
  float bound = argc>1? atof(argv[1]) : 1000000.0f;
  uint32_t ub = f2u(bound);
  long hist[2][4]={{0}}; 
  double maxerr[2]={0,0}; uint32_t worst[2]={0,0};
  long zero_bad=0;
  #pragma omp parallel
  {
    long h[2][4]={{0}}; double me[2]={0,0}; uint32_t w[2]={0,0}; long zb=0;
    #pragma omp for schedule(dynamic, 1<<16)
    for (uint32_t i=0;i<ub;i++){
      for (int sgn=0;sgn<2;sgn++){
        float x=u2f(i|((uint32_t)sgn<<31));
        for (int c=0;c<2;c++){
          double tr = c? cos((double)x): sin((double)x);
          float o=(float)tr; float g=sincos_f32(x,c);
          int32_t a=(int32_t)f2u(o), b=(int32_t)f2u(g);
          long d;
          if ((a^b)<0) { d = (o==g)? ( (f2u(o)!=f2u(g)) ? (zb++,0):0) : 3; }
          else d=labs((long)a-(long)b);
          if (d>3) d=3;
          h[c][d]++;
          double ulp = ldexp(1.0, ilogb(tr==0?1e-300:tr)-23);
          if (fabs(tr) < 1.1754943508222875e-38) ulp = ldexp(1.0,-149);
          double er=fabs((double)g-tr)/ulp;
          if (er>me[c]){me[c]=er;w[c]=f2u(x);}
        }
      }
    }
    #pragma omp critical
    { for(int c=0;c<2;c++){for(int k=0;k<4;k++)hist[c][k]+=h[c][k]; if(me[c]>maxerr[c]){maxerr[c]=me[c];worst[c]=w[c];}} zero_bad+=zb; }
  }
  for(int c=0;c<2;c++) printf("%s: 0ulp %ld 1ulp %ld 2ulp %ld >=3 %ld  max err vs true %.4f ULP at x=%a (0x%08x)\n", c?"cos":"sin", hist[c][0],hist[c][1],hist[c][2],hist[c][3],maxerr[c],u2f(worst[c]),worst[c]);
  printf("signed-zero mismatches: %ld\n", zero_bad);
  return 0;
}
