// DEV TOOL: issue rate of the VALU / LDS instructions the f32 pow / log / sinh kernels are made of, on gfx950.
// Each kernel runs ITER × 8 independent copies of ONE instruction per wave with 8 waves per SIMD resident; the time
// relative to v_fma_f32 is the instruction's cost in "f32-fma units" — the unit DESIGN.md §4 prices pow in.
//   hipcc -O2 --offload-arch=gfx950 tools/probe/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum Op { FMA_F32, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32, FMA_F64, MUL_F64, ADD_F64, CVT_F32_F64, CVT_F64_F32, CVT_F64_I32, CVT_I32_F64, RNDNE_F64,
          LDEXP_F64, LDEXP_F32, AND_B32, BFE_U32, CNDMASK, CMP_LT_U32, ADD3_U32, LSHL_OR, RCP_F32, EXP_F32, LOG_F32, RNDNE_F32, CVT_F32_I32, CVT_I32_F32,
          MUL_LO_U32, DS_READ_B128, DS_READ_B64, FREXP_MANT_F64, RCP_F64, CVT_F32_U32, MAX_F64, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f32_f64",
                                    "v_cvt_f64_f32", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_rndne_f64", "v_ldexp_f64", "v_ldexp_f32", "v_and_b32",
                                    "v_bfe_u32", "v_cndmask_b32", "v_cmp_lt_u32", "v_add3_u32", "v_lshl_or_b32", "v_rcp_f32", "v_exp_f32",
                                    "v_log_f32", "v_rndne_f32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_mul_lo_u32", "ds_read_b128", "ds_read_b64",
                                    "v_frexp_mant_f64", "v_rcp_f64", "v_cvt_f32_u32", "v_max_f64"};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters) {
  __shared__ u32x4 lds[256];
  lds[threadIdx.x] = u32x4{threadIdx.x, 1u, 2u, 3u};
  __syncthreads();
  float f[8];
  double d[8];
  int32_t i[8];
  u32x4 q[8];
  u32x2 h[8];
  const float fa = 1.0f + 1e-7f * threadIdx.x, fb = 1e-9f;
  const double da = 1.0 + 1e-12 * threadIdx.x, db = 1e-18;
  const int32_t ia = 0x00ffff0f, ib = 3;
  const uint32_t addr = (threadIdx.x & 127u) * 16u;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    f[r] = 1.0f + r + threadIdx.x;
    d[r] = 1.0 + r + threadIdx.x;
    i[r] = r + threadIdx.x;
    q[r] = u32x4{0, 0, 0, 0};
    h[r] = u32x2{0, 0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(fa), "v"(fb));
      if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[r]) : "v"(da), "v"(db));
      if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[r]) : "v"(da));
      if constexpr (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[r]) : "v"(da));
      if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[r]) : "v"(da), "v"(db));
      if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[r]) : "v"(da));
      if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[r]) : "v"(db));
      if constexpr (OP == MAX_F64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[r]) : "v"(db));
      if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[r]) : "v"(d[r]));
      if constexpr (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[r]) : "v"(f[r]));
      if constexpr (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[r]) : "v"(i[r]));
      if constexpr (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i[r]) : "v"(d[r]));
      if constexpr (OP == RNDNE_F64) asm volatile("v_rndne_f64 %0, %1" : "=v"(d[r]) : "v"(d[r]));
      if constexpr (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[r]) : "v"(ib));
      if constexpr (OP == LDEXP_F32) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[r]) : "v"(ib));
      if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i[r]) : "v"(ia));
      if constexpr (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(i[r]));
      if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[r]) : "v"(ia));
      if constexpr (OP == CMP_LT_U32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(i[r]), "v"(ia) : "vcc");
      if constexpr (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(i[r]) : "v"(ia), "v"(ib));
      if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(i[r]) : "v"(ib), "v"(ia));
      if constexpr (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[r]));
      if constexpr (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(f[r]));
      if constexpr (OP == LOG_F32) asm volatile("v_log_f32 %0, %0" : "+v"(f[r]));
      if constexpr (OP == RNDNE_F32) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[r]));
      if constexpr (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[r]) : "v"(i[r]));
      if constexpr (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[r]) : "v"(i[r]));
      if constexpr (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i[r]) : "v"(f[r]));
      if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i[r]) : "v"(ib));
      if constexpr (OP == FREXP_MANT_F64) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[r]));
      if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[r]));
      if constexpr (OP == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(q[r]) : "v"(addr));
      if constexpr (OP == DS_READ_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(h[r]) : "v"(addr));
    }
    if constexpr (OP == DS_READ_B128 || OP == DS_READ_B64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < 8; r++)
    acc += __builtin_bit_cast(uint32_t, f[r]) + (uint32_t)__builtin_bit_cast(uint64_t, d[r]) + (uint32_t)i[r] + q[r].x + h[r].x;
  out[(uint64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
static int run_one(uint32_t* out, int blocks, int iters, double* ms_out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(256), 0, nullptr, out, iters);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  *ms_out = ts[ts.size() / 2];
  return 0;
}

template <int OP>
static int run_all(uint32_t* out, int blocks, int iters, double* ms) {
  if constexpr (OP < N_OPS) {
    if (run_one<OP>(out, blocks, iters, &ms[OP])) return 1;
    return run_all<OP + 1>(out, blocks, iters, ms);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks × 4 waves per CU = 8 waves per SIMD
  const int iters = 20000;
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  double ms[N_OPS];
  if (run_all<0>(out, blocks, iters, ms)) return 1;
  // wave-instructions per SIMD: 8 waves × iters × 8
  const double per_simd = 8.0 * iters * 8.0;
  printf("{\"cus\": %d, \"clock_mhz_reported\": %d, \"iters\": %d, \"rows\": [\n", cus, prop.clockRate / 1000, iters);
  for (int o = 0; o < N_OPS; o++)
    printf("  {\"op\": \"%s\", \"ms\": %.4f, \"ns_per_wave_instr\": %.3f, \"units_vs_fma_f32\": %.2f}%s\n", kNames[o], ms[o],
           ms[o] * 1e6 / per_simd, ms[o] / ms[FMA_F32], o + 1 < N_OPS ? "," : "");
  printf("]}\n");
  CK(hipFree(out));
  return 0;
}
