// DEV TOOL: what bounds the reference-order f32 tree sum (reduce.hip sum_tree_span_kernel, 0.77 of the HBM roof against
// 0.82–0.86 for the order-free reductions)?  Same load pattern, different amounts of cross-lane work and block shapes.
//   variant 0: the product's shape — 256-thread block per 65536-row span, 8 blocks per step, transpose-reduce
//   variant 1: same loads, NO cross-lane work (plain per-lane accumulation; wrong order — the ceiling of the access pattern)
//   variant 2: one WAVE per block, one 16384-row quarter span per wave (no LDS, no barrier); partial per wave
//   variant 3: as 2 with 4 blocks per step instead of 8
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_tree_sum(float v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) v = v + __shfl_down(v, off);
  return v;
}
__device__ __forceinline__ float transpose_reduce8(const float* s, uint32_t lane) {
  float t4[4], t2[2];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool odd = lane & 1;
    const float keep = odd ? s[2 * j + 1] : s[2 * j];
    const float send = odd ? s[2 * j] : s[2 * j + 1];
    t4[j] = keep + __shfl_xor(send, 1);
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const bool odd = lane & 2;
    const float keep = odd ? t4[2 * j + 1] : t4[2 * j];
    const float send = odd ? t4[2 * j] : t4[2 * j + 1];
    t2[j] = keep + __shfl_xor(send, 2);
  }
  float v;
  {
    const bool odd = lane & 4;
    const float keep = odd ? t2[1] : t2[0];
    const float send = odd ? t2[0] : t2[1];
    v = keep + __shfl_xor(send, 4);
  }
  v = v + __shfl_xor(v, 8);
  v = v + __shfl_xor(v, 16);
  v = v + __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float transpose_reduce4(const float* s, uint32_t lane) {
  float t2[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const bool odd = lane & 1;
    const float keep = odd ? s[2 * j + 1] : s[2 * j];
    const float send = odd ? s[2 * j] : s[2 * j + 1];
    t2[j] = keep + __shfl_xor(send, 1);
  }
  float v;
  {
    const bool odd = lane & 2;
    const float keep = odd ? t2[1] : t2[0];
    const float send = odd ? t2[0] : t2[1];
    v = keep + __shfl_xor(send, 2);
  }
  v = v + __shfl_xor(v, 4);
  v = v + __shfl_xor(v, 8);
  v = v + __shfl_xor(v, 16);
  v = v + __shfl_xor(v, 32);
  return v;  // every lane: the sum of block (lane & 3)
}

template <int VARIANT>
__global__ __launch_bounds__(256) void span_kernel(const float* in, uint64_t nspans, float* partials) {
  __shared__ float lds[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t sp = blockIdx.x; sp < nspans; sp += gridDim.x) {
    const uint64_t wave_base = sp * 65536 + (uint64_t)wave * 16384;
    float acc = 0.0f;
    for (int j0 = 0; j0 < 64; j0 += 8) {
      float s[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + wave_base + (uint64_t)(j0 + u) * 256 + lane * 4));
        s[u] = (v.x + v.y) + (v.z + v.w);
      }
      if (VARIANT == 1) {
        acc += ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
      } else {
        const float v = transpose_reduce8(s, lane);
        if ((lane >> 3) == (uint32_t)(j0 >> 3)) acc = v;
      }
    }
    const float wsum = wave_tree_sum(acc);
    __syncthreads();
    if (lane == 0) lds[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[sp] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  }
}

template <int UNR>
__global__ __launch_bounds__(64) void wave_kernel(const float* in, uint64_t nquarters, float* partials) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t qd = blockIdx.x; qd < nquarters; qd += gridDim.x) {
    const uint64_t wave_base = qd * 16384;
    float acc = 0.0f;
    for (int j0 = 0; j0 < 64; j0 += UNR) {
      float s[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + wave_base + (uint64_t)(j0 + u) * 256 + lane * 4));
        s[u] = (v.x + v.y) + (v.z + v.w);
      }
      if (UNR == 8) {
        const float v = transpose_reduce8(s, lane);
        if ((lane >> 3) == (uint32_t)(j0 >> 3)) acc = v;
      } else {
        const float v = transpose_reduce4(s, lane);
        if ((lane >> 2) == (uint32_t)(j0 >> 2)) acc = v;
      }
    }
    const float wsum = wave_tree_sum(acc);
    if (lane == 0) partials[qd] = wsum;
  }
}

extern "C" int probe_sum(const void* in, uint64_t n, void* partials, int variant, int grid_cap, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nspans = n / 65536, nq = n / 16384;
  const float* pi = (const float*)in;
  float* po = (float*)partials;
  if (variant == 0 || variant == 1) {
    const unsigned grid = (unsigned)(grid_cap > 0 && (uint64_t)grid_cap < nspans ? (uint64_t)grid_cap : nspans);
    if (variant == 0) hipLaunchKernelGGL(span_kernel<0>, dim3(grid), dim3(256), 0, s, pi, nspans, po);
    else hipLaunchKernelGGL(span_kernel<1>, dim3(grid), dim3(256), 0, s, pi, nspans, po);
  } else {
    const unsigned grid = (unsigned)(grid_cap > 0 && (uint64_t)grid_cap < nq ? (uint64_t)grid_cap : nq);
    if (variant == 2) hipLaunchKernelGGL(wave_kernel<8>, dim3(grid), dim3(64), 0, s, pi, nq, po);
    else hipLaunchKernelGGL(wave_kernel<4>, dim3(grid), dim3(64), 0, s, pi, nq, po);
  }
  return (int)hipGetLastError();
}
