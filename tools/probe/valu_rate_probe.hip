// valu_rate_probe.hip — issue rate of the VALU instructions the f32 pow / sin / log kernels are made of, on gfx950:
// every workgroup runs ITERS × 8 independent chains of ONE instruction kind, nothing touches memory.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate_probe.hip -o valu_rate_probe && ./valu_rate_probe
// Output: one JSON line per instruction: G lane-ops/s chip-wide and the rate relative to v_fma_f32.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define ITERS 4096
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(double* sink, float seed_f, double seed_d, int seed_i) {
  float f[CHAINS];
  double d[CHAINS];
  int n[CHAINS];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) {
    f[c] = seed_f + c + threadIdx.x * 1e-3f;
    d[c] = seed_d + c + threadIdx.x * 1e-3;
    n[c] = seed_i + c;
    p[c] = v2f{f[c], f[c] + 1.0f};
  }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if constexpr (KIND == 0) f[c] = __builtin_fmaf(f[c], 1.0000001f, 1e-7f);
      else if constexpr (KIND == 1) d[c] = __builtin_fma(d[c], 1.0000001, 1e-9);
      else if constexpr (KIND == 2) d[c] = d[c] * 1.0000001;
      else if constexpr (KIND == 3) d[c] = d[c] + 1e-9;
      else if constexpr (KIND == 4) { d[c] = (double)f[c]; f[c] = f[c] + (float)(it & 1); asm volatile("" : "+v"(d[c])); }   // cvt_f64_f32 (+1 f32 add)
      else if constexpr (KIND == 5) { f[c] = (float)d[c]; d[c] = d[c] + 1e-9; asm volatile("" : "+v"(f[c])); }               // cvt_f32_f64 (+1 f64 add)
      else if constexpr (KIND == 6) d[c] = __builtin_rint(d[c] * 1.5);                                                       // rndne_f64 (+1 f64 mul)
      else if constexpr (KIND == 7) d[c] = __builtin_ldexp(d[c], n[c] & 1);                                                  // ldexp_f64 (+1 and)
      else if constexpr (KIND == 8) { n[c] = (int)d[c]; d[c] = d[c] + 1e-9; asm volatile("" : "+v"(n[c])); }                 // cvt_i32_f64 (+1 f64 add)
      else if constexpr (KIND == 9) { d[c] = (double)n[c]; n[c] = n[c] + 1; asm volatile("" : "+v"(d[c])); }                 // cvt_f64_i32 (+1 int add)
      else if constexpr (KIND == 10) p[c] = __builtin_elementwise_fma(p[c], v2f{1.0000001f, 1.0000001f}, v2f{1e-7f, 1e-7f}); // v_pk_fma_f32
      else if constexpr (KIND == 11) d[c] = d[c] < 2000.0 ? d[c] : 2000.0;                                                   // min_f64
      else if constexpr (KIND == 12) f[c] = f[c] + 1e-7f;                                                                    // add_f32
      else if constexpr (KIND == 13) n[c] = (n[c] >> 3) + (n[c] << 5);                                                       // 3 int ops
    }
  }
  double acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc += f[c] + d[c] + n[c] + p[c].x + p[c].y;
  if (acc == 123.456) sink[0] = acc;
}

template <int KIND>
static void run(const char* name, double ops_per_iter_chain, double* sink, double* base) {
  const int grid = 256 * 8;
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((rate_kernel<KIND>), dim3(grid), dim3(256), 0, 0, sink, 1.0f, 1.0, 3);
  hipDeviceSynchronize();
  hipEventRecord(s, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((rate_kernel<KIND>), dim3(grid), dim3(256), 0, 0, sink, 1.0f, 1.0, 3);
  hipEventRecord(e, 0);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  const double slots = 5.0 * grid * 256.0 * ITERS * CHAINS;  // (iteration, chain, lane) slots executed
  const double gslots = slots / (ms * 1e-3) / 1e9;
  if (KIND == 0) *base = gslots;
  printf("{\"instr\": \"%s\", \"ms\": %.3f, \"G_slots_per_s\": %.1f, \"instrs_per_slot\": %.0f, \"slot_cost_vs_fma_f32\": %.2f}\n", name, ms / 5,
         gslots, ops_per_iter_chain, *base / gslots);
}

int main() {
  double* sink;
  hipMalloc(&sink, 64);
  double base = 1;
  run<0>("v_fma_f32", 1, sink, &base);
  run<12>("v_add_f32", 1, sink, &base);
  run<10>("v_pk_fma_f32 (2 rows per instr)", 1, sink, &base);
  run<1>("v_fma_f64", 1, sink, &base);
  run<2>("v_mul_f64", 1, sink, &base);
  run<3>("v_add_f64", 1, sink, &base);
  run<11>("v_min_f64 (cmp+cndmask or min)", 1, sink, &base);
  run<4>("v_cvt_f64_f32 + v_add_f32", 2, sink, &base);
  run<5>("v_cvt_f32_f64 + v_add_f64", 2, sink, &base);
  run<6>("v_rndne_f64 + v_mul_f64", 2, sink, &base);
  run<7>("v_ldexp_f64 + v_and", 2, sink, &base);
  run<8>("v_cvt_i32_f64 + v_add_f64", 2, sink, &base);
  run<9>("v_cvt_f64_i32 + v_add_i32", 2, sink, &base);
  run<13>("3 integer ops (2 shifts + add)", 3, sink, &base);
  return 0;
}
