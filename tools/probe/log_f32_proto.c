// DEV TOOL (round 6): CPU emulation of the packed-f32 log of csrc/elementwise.hip (log_f32_pair) — the SAME operation sequence, fmaf = one rounding —
// against f64 libm rounded once (the oracle) for EVERY positive normal f32: histogram of ULP distances, largest error against the true value.
// Coefficients: tools/probe/log_f32_fit.py.   gcc -O2 -march=native -ffp-contract=off -fopenmp -DNC=8 tools/probe/log_f32_proto.c -o /tmp/lp -lm && /tmp/lp
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <omp.h>
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
#ifndef NC
#define NC 8  // the shipped form: eight coefficients
#endif
static const float P9[9]={0x1.555548p-2f,-0x1.000006p-2f,0x1.99a4b6p-3f,-0x1.555c4ep-3f,0x1.233768p-3f,-0x1.fc2208p-4f,0x1.e6f06ap-4f,-0x1.de3c16p-4f,0x1.1457aap-4f};
static const float P8[8]={0x1.555554p-2f,-0x1.000226p-2f,0x1.99a008p-3f,-0x1.547244p-3f,0x1.22da1cp-3f,-0x1.0d8542p-3f,0x1.055b6cp-3f,-0x1.38b586p-4f};
static inline float log_f32(float x){  // positive normal x
  const uint32_t xb=f2u(x);
  const uint32_t s = xb - 0x3f3504f3u;            // m in [sqrt(1/2), sqrt(2))
  const int32_t e = (int32_t)s >> 23;
  const float m = u2f(xb - ((uint32_t)e << 23));
  const float ef=(float)e;
  const float f = m - 1.0f;
  const float f2 = f*f;
  const float *P = NC==9?P9:P8;
  float p = P[NC-1];
  for(int k=NC-2;k>=0;k--) p=fmaf(f,p,P[k]);
  const float f3 = f2*f;
  const float h = 0.5f*f2;                 // exact scaling
  const float q = fmaf(f3,p,-h);           // f^3 P - f^2/2
  const float r = f + q;
  const float LN2_HI = 0x1.62e4p-1f, LN2_LO = 0x1.7f7d1cp-20f; // hi has 15 significant bits
  const float t = fmaf(ef, LN2_LO, r);
  return fmaf(ef, LN2_HI, t);
}
int main(){
  long hist[4]={0}; double me=0; uint32_t w=0;
  #pragma omp parallel
  {
    long h[4]={0}; double m_=0; uint32_t w_=0;
    #pragma omp for schedule(dynamic,1<<16)
    for(uint32_t i=0x00800000u;i<0x7f800000u;i++){
      float x=u2f(i); double tr=log((double)x); float o=(float)tr, g=log_f32(x);
      int32_t a=(int32_t)f2u(o), b=(int32_t)f2u(g);
      long d = ((a^b)<0) ? ((o==g)?0:3) : labs((long)a-(long)b); if(d>3)d=3; h[d]++;
      if (tr!=0){ double ulp=ldexp(1.0,ilogb(tr)-23); double er=fabs((double)g-tr)/ulp; if(er>m_){m_=er;w_=i;} }
    }
    #pragma omp critical
    { for(int k=0;k<4;k++)hist[k]+=h[k]; if(m_>me){me=m_;w=w_;} }
  }
  printf("NC=%d: 0ulp %ld 1ulp %ld 2ulp %ld >=3 %ld max err %.4f at %a (0x%08x)\n",NC,hist[0],hist[1],hist[2],hist[3],me,u2f(w),w);
}
