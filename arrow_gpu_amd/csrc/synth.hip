// synth.hip — counter-based synthetic columns and an order-independent checksum (bench / full-size parity only).
//
// Not reference functionality: the reference builds inputs on the host with from_slice
// (crates/array/src/array/primitive_array_gpu.rs:55-66); 1e9-row columns are generated in HBM instead so the
// bench never moves 12 GB over PCIe.  Element i depends only on (seed, row0+i), and the CPU oracle
// (oracle/agpu_oracle.c synth_*) regenerates any window bit-identically.
#include "common.hpp"

__device__ __forceinline__ float synth_f32_one(uint64_t seed, uint64_t row, float lo, float hi) {
  const float u = __fmul_rn((float)(uint32_t)(row_hash_dev(seed, row) >> 40), 0x1p-24f);
  return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), u));
}

__global__ __launch_bounds__(AGPU_BLOCK) void synth_f32_kernel(float* out, uint64_t n, uint64_t seed, uint64_t row0,
                                                              float lo, float hi, int vec_ok) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t npacks = vec_ok ? n / 4 : 0;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const uint64_t r = row0 + pk * 4;
    f32x4 v = {synth_f32_one(seed, r, lo, hi), synth_f32_one(seed, r + 1, lo, hi), synth_f32_one(seed, r + 2, lo, hi),
               synth_f32_one(seed, r + 3, lo, hi)};
    *reinterpret_cast<f32x4*>(out + pk * 4) = v;
  }
  for (uint64_t i = npacks * 4 + tid; i < n; i += stride) out[i] = synth_f32_one(seed, row0 + i, lo, hi);
}

__device__ __forceinline__ uint32_t synth_i32_one(uint64_t seed, uint64_t row, uint32_t modulus) {
  const uint32_t v = (uint32_t)(row_hash_dev(seed, row) >> 32);
  return modulus ? v % modulus : v;
}

__global__ __launch_bounds__(AGPU_BLOCK) void synth_i32_kernel(uint32_t* out, uint64_t n, uint64_t seed, uint64_t row0,
                                                              uint32_t modulus, int vec_ok) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t npacks = vec_ok ? n / 4 : 0;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const uint64_t r = row0 + pk * 4;
    u32x4 v = {synth_i32_one(seed, r, modulus), synth_i32_one(seed, r + 1, modulus), synth_i32_one(seed, r + 2, modulus),
               synth_i32_one(seed, r + 3, modulus)};
    *reinterpret_cast<u32x4*>(out + pk * 4) = v;
  }
  for (uint64_t i = npacks * 4 + tid; i < n; i += stride) out[i] = synth_i32_one(seed, row0 + i, modulus);
}

__global__ __launch_bounds__(AGPU_BLOCK) void synth_u8_kernel(uint8_t* out, uint64_t n, uint64_t seed, uint64_t row0,
                                                             int vec_ok) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t npacks = vec_ok ? n / 16 : 0;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const uint64_t r = row0 + pk * 16;
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint32_t x = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) x |= (uint32_t)(row_hash_dev(seed, r + q * 4 + k) >> 56) << (8 * k);
      w[q] = x;
    }
    u32x4 v = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4*>(out + pk * 16) = v;
  }
  for (uint64_t i = npacks * 16 + tid; i < n; i += stride) out[i] = (uint8_t)(row_hash_dev(seed, row0 + i) >> 56);
}

// one 64-bit word per wave round via ballot; padding bits zero
__global__ __launch_bounds__(AGPU_BLOCK) void synth_bits_kernel(uint64_t* out, uint64_t n_bits, uint64_t seed,
                                                               uint64_t row0, double p_set) {
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  const uint64_t wave_id = ((uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x) / AGPU_WAVE;
  const uint64_t n_waves = (uint64_t)gridDim.x * (AGPU_BLOCK / AGPU_WAVE);
  const uint64_t nwords = (n_bits + 63) / 64;
  for (uint64_t w = wave_id; w < nwords; w += n_waves) {
    const uint64_t i = w * 64 + lane;
    bool bit = false;
    if (i < n_bits) {
      const double u = __dmul_rn((double)(row_hash_dev(seed, row0 + i) >> 11), 0x1p-53);
      bit = u < p_set;
    }
    const uint64_t m = __ballot(bit);
    if (lane == 0) out[w] = m;
  }
}

__global__ __launch_bounds__(AGPU_BLOCK) void checksum_kernel(const uint8_t* data, uint64_t bytes,
                                                             unsigned long long* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t nw = bytes / 8;
  uint64_t s = 0;
  const uint64_t* d64 = reinterpret_cast<const uint64_t*>(data);
  for (uint64_t k = tid; k < nw; k += stride) s += splitmix64_dev(d64[k] ^ k);
  if (tid == 0 && (bytes & 7)) {
    uint64_t w = 0;
    for (uint32_t j = 0; j < (bytes & 7); j++) w |= (uint64_t)data[nw * 8 + j] << (8 * j);
    s += splitmix64_dev(w ^ nw);
  }
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & (AGPU_WAVE - 1)) == 0 && s) atomicAdd(out, (unsigned long long)s);
}

extern "C" {

agpu_status agpu_synth_f32(agpu_pipeline* p, float* out, uint64_t n, uint64_t seed, uint64_t row0, float lo, float hi) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(out, AGPU_ERR_ARG, "null pointer");
  const int grid = stream_grid_for(p, (n / 4 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(synth_f32_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, out, n, seed, row0, lo, hi,
                     aligned16(out) ? 1 : 0);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_synth_i32(agpu_pipeline* p, int32_t* out, uint64_t n, uint64_t seed, uint64_t row0, uint32_t modulus) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(out, AGPU_ERR_ARG, "null pointer");
  const int grid = stream_grid_for(p, (n / 4 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(synth_i32_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, reinterpret_cast<uint32_t*>(out), n,
                     seed, row0, modulus, aligned16(out) ? 1 : 0);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_synth_u8(agpu_pipeline* p, uint8_t* out, uint64_t n, uint64_t seed, uint64_t row0) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(out, AGPU_ERR_ARG, "null pointer");
  const int grid = stream_grid_for(p, (n / 16 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(synth_u8_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, out, n, seed, row0,
                     aligned16(out) ? 1 : 0);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_synth_bits(agpu_pipeline* p, void* out_bits, uint64_t n_bits, uint64_t seed, uint64_t row0,
                            double p_set) {
  AGPU_BIND(p);
  if (n_bits == 0) return AGPU_OK;
  AGPU_REQUIRE(out_bits && aligned_to(out_bits, 8), AGPU_ERR_SHAPE, "bitmap must be 8-byte aligned");
  const uint64_t nwords = (n_bits + 63) / 64;
  const int grid = stream_grid_for(p, (nwords + 3) / 4);
  hipLaunchKernelGGL(synth_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<uint64_t*>(out_bits),
                     n_bits, seed, row0, p_set);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_checksum(agpu_pipeline* p, const void* data, uint64_t bytes, uint64_t* out_sum_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_sum_dev, AGPU_ERR_ARG, "null output");
  AGPU_HIP(hipMemsetAsync(out_sum_dev, 0, sizeof(uint64_t), p->stream));
  if (bytes == 0) return AGPU_OK;
  AGPU_REQUIRE(data && aligned_to(data, 8), AGPU_ERR_SHAPE, "data must be 8-byte aligned");
  const int grid = atomic_grid_for(p, (bytes / 8 + AGPU_BLOCK * 4 - 1) / (AGPU_BLOCK * 4));  // same-address atomics
  hipLaunchKernelGGL(checksum_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint8_t*>(data),
                     bytes, reinterpret_cast<unsigned long long*>(out_sum_dev));
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

}  // extern "C"
