// DEV TOOL: shapes for the 1-byte → 4-byte widening stream (cast u8→f32: 1 B read + 4 B written per row).
//   variant 0: lane loads 4 B (4 rows), stores one 16-byte vector            (what cvt_kernel does)
//   variant 1: lane loads 16 B (16 rows); 4 ds_bpermute redistribute the words so that each of the 4 stores of the
//              wave is a fully coalesced 1 KiB row (lane l of store j writes rows 256j + 4l .. +3)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 cvt4(uint32_t w) {
  f32x4 r = {(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
  return r;
}

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void cast_v0(const uint32_t* in, f32x4* out, uint64_t ntiles) {
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * (uint64_t)(BLOCK * U) + threadIdx.x;
    uint32_t w[U];
#pragma unroll
    for (int u = 0; u < U; u++) w[u] = __builtin_nontemporal_load(in + p0 + (uint64_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(cvt4(w[u]), out + p0 + (uint64_t)u * BLOCK);
  }
}

// one wave handles 1024 rows per step: 1 KiB in, 4 KiB out
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void cast_v1(const u32x4* in, f32x4* out, uint64_t ntiles) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WAVES = BLOCK / 64;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t chunk = (t * U + u) * WAVES + wave;  // 1024-row chunk index
      const u32x4 v = __builtin_nontemporal_load(in + chunk * 64 + lane);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // store j, lane l needs word (l & 3) of source lane 16 j + (l >> 2)
        const int src = (16 * j + (int)(lane >> 2)) * 4;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.x);
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.y);
        const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.z);
        const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.w);
        const uint32_t sel = lane & 3;
        const uint32_t w = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
        __builtin_nontemporal_store(cvt4(w), out + chunk * 256 + j * 64 + lane);
      }
    }
  }
}

// variant 2: the product's shape (cvt_wide_kernel): 16 B per lane in, transposed through 1 KiB of LDS, 4 coalesced stores
template <int U>
__global__ __launch_bounds__(64) void cast_v2(const u32x4* in, f32x4* out, uint64_t nchunks) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[U][64 * 16];
  const uint32_t lane = threadIdx.x;
  for (uint64_t c0 = (uint64_t)blockIdx.x * U; c0 < nchunks; c0 += (uint64_t)gridDim.x * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(in + (c0 + u) * 64 + lane);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) *reinterpret_cast<u32x4*>(lds[u] + lane * 16) = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t g = j * 64 + lane;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(lds[u] + g * 4);
        __builtin_nontemporal_store(cvt4(w), out + (c0 + u) * 256 + g);
      }
  }
}
// variant 3: as 2 but plain (temporal) stores
__global__ __launch_bounds__(64) void cast_v3(const u32x4* in, f32x4* out, uint64_t nchunks) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 16];
  const uint32_t lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const u32x4 v = __builtin_nontemporal_load(in + c * 64 + lane);
    __syncthreads();
    *reinterpret_cast<u32x4*>(lds + lane * 16) = v;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t g = j * 64 + lane;
      const uint32_t w = *reinterpret_cast<const uint32_t*>(lds + g * 4);
      out[c * 256 + g] = cvt4(w);
    }
  }
}

// variant 4: as 1, but the transposition costs 4 ds_permute (push) instead of 16 ds_bpermute: in round d every lane s pushes
// its dword (d + s/16) & 3 to lane 4 (s % 16) + ((d + s/16) & 3) — a permutation of the 64 lanes — so lane l receives, in
// round d, the dword of store j = ((l & 3) - d) & 3; two 4-way selects per lane replace the other 12 crossbar trips
__global__ __launch_bounds__(64) void cast_v4(const u32x4* in, f32x4* out, uint64_t nchunks) {
  const uint32_t lane = threadIdx.x;
  const uint32_t sj = lane >> 4, st = lane & 15, e = lane & 3;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const u32x4 v = __builtin_nontemporal_load(in + c * 64 + lane);
    uint32_t r[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t k = (d + sj) & 3;
      const uint32_t val = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
      r[d] = (uint32_t)__builtin_amdgcn_ds_permute((int)((4 * st + k) * 4), (int)val);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t d = (e - j) & 3;
      const uint32_t w = d == 0 ? r[0] : d == 1 ? r[1] : d == 2 ? r[2] : r[3];
      __builtin_nontemporal_store(cvt4(w), out + c * 256 + j * 64 + lane);
    }
  }
}

// variant 5: as 1 with PLAIN (temporal) stores — a pure store stream measured faster without `nt` (memset_check.py)
__global__ __launch_bounds__(64) void cast_v5(const u32x4* in, f32x4* out, uint64_t nchunks) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const u32x4 v = __builtin_nontemporal_load(in + c * 64 + lane);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int src = (16 * j + (int)(lane >> 2)) * 4;
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.x);
      const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.y);
      const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.z);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.w);
      const uint32_t sel = lane & 3;
      const uint32_t w = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
      out[c * 256 + j * 64 + lane] = cvt4(w);
    }
  }
}


// variant 6 (round 3): LOAD-WAVE / STORE-WAVE split through LDS — the idea DESIGN listed as untried.  A block of (1 + SW) waves:
// wave 0 only LOADS (K chunks of 1 KiB = K 16-byte loads per lane per round, a K KiB read burst) into one half of a double
// buffer in LDS; waves 1..SW only STORE: wave w converts chunks w-1, w-1+SW, … of the round and writes each as four coalesced
// 1 KiB rows (the dword of store j, lane l sits at LDS word j*64 + l of the chunk: conflict-free reads, no crossbar trips).
// Persistent over the grid; one barrier per round.
template <int SW, int K>
__global__ __launch_bounds__(64 * (1 + SW)) void cast_v6(const u32x4* in, f32x4* out, uint64_t nrounds) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][K * 256];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t r = blockIdx.x;
  if (wave == 0 && r < nrounds) {
#pragma unroll
    for (int k = 0; k < K; k++) *reinterpret_cast<u32x4*>(&lds[0][k * 256 + lane * 4]) = __builtin_nontemporal_load(in + (r * K + k) * 64 + lane);
  }
  __syncthreads();
  int buf = 0;
  while (r < nrounds) {
    const uint64_t nxt = r + gridDim.x;
    if (wave == 0) {
      if (nxt < nrounds) {
#pragma unroll
        for (int k = 0; k < K; k++) *reinterpret_cast<u32x4*>(&lds[buf ^ 1][k * 256 + lane * 4]) = __builtin_nontemporal_load(in + (nxt * K + k) * 64 + lane);
      }
    } else {
      for (int k = (int)wave - 1; k < K; k += SW) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t w = lds[buf][k * 256 + j * 64 + lane];
          __builtin_nontemporal_store(cvt4(w), out + (r * K + k) * 256 + j * 64 + lane);
        }
      }
    }
    __syncthreads();
    buf ^= 1;
    r = nxt;
  }
}

// variant 7: no dedicated waves, but the READS of a 256-thread block go out as one burst first: every wave loads its chunk, all
// four chunks meet in LDS, then every wave stores its own chunk's four rows from LDS (conflict-free) — cvt_wide's shape with the
// transposition done by LDS memory instead of the crossbar, one block-wide barrier per tile
__global__ __launch_bounds__(256) void cast_v7(const u32x4* in, f32x4* out, uint64_t ntiles) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 256];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const u32x4 v = __builtin_nontemporal_load(in + (t * 4 + wave) * 64 + lane);
    __syncthreads();
    *reinterpret_cast<u32x4*>(&lds[wave * 256 + lane * 4]) = v;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t w = lds[wave * 256 + j * 64 + lane];
      __builtin_nontemporal_store(cvt4(w), out + (t * 4 + wave) * 256 + j * 64 + lane);
    }
  }
}


// variant 8 (round 3): the STORE shape that a pure fill likes best — 256 threads, ONE 16-byte store per lane (store_probe: 0.85
// against 0.80 for one wave x 4 stores) — fed by ONE wave's 16-byte loads through 1 KiB of LDS: a block owns one 1 KiB chunk
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void cast_v8(const u32x4* in, f32x4* out, uint64_t nchunks) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[256];
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    if (threadIdx.x < 64) *reinterpret_cast<u32x4*>(&lds[threadIdx.x * 4]) = __builtin_nontemporal_load(in + c * 64 + threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 256 / BLOCK; j++) {
      const uint32_t g = j * BLOCK + threadIdx.x;
      __builtin_nontemporal_store(cvt4(lds[g]), out + c * 256 + g);
    }
    if (gridDim.x < nchunks) __syncthreads();
  }
}

extern "C" int probe_cast(const void* in, void* out, uint64_t n, int variant, int block, int u, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define GO0(B, U_)                                                                                         \
  {                                                                                                        \
    uint64_t nt = n / 4 / ((uint64_t)B * U_);                                                              \
    hipLaunchKernelGGL((cast_v0<B, U_>), dim3((unsigned)nt), dim3(B), 0, s, (const uint32_t*)in, (f32x4*)out, nt); \
  }
#define GO1(B, U_)                                                                                         \
  {                                                                                                        \
    uint64_t nt = n / 1024 / ((uint64_t)(B / 64) * U_);                                                    \
    hipLaunchKernelGGL((cast_v1<B, U_>), dim3((unsigned)nt), dim3(B), 0, s, (const u32x4*)in, (f32x4*)out, nt); \
  }
  if (variant == 6) {  // block = store waves (1, 2 or 4), u = chunks per round (4 or 8); grid = persistent (8 blocks per CU)
    const int sw = block, k = u;
    const uint64_t nrounds = n / 1024 / (uint64_t)k;
    const unsigned grid = (unsigned)(nrounds < 2048 ? nrounds : 2048);
#define GO6(SW, K) hipLaunchKernelGGL((cast_v6<SW, K>), dim3(grid), dim3(64 * (1 + SW)), 0, s, (const u32x4*)in, (f32x4*)out, nrounds)
    if (sw == 1 && k == 4) GO6(1, 4); else if (sw == 2 && k == 4) GO6(2, 4); else if (sw == 4 && k == 4) GO6(4, 4);
    else if (sw == 2 && k == 8) GO6(2, 8); else if (sw == 4 && k == 8) GO6(4, 8); else if (sw == 4 && k == 16) GO6(4, 16); else return 1;
    return (int)hipGetLastError();
  }
  if (variant == 8) {
    const uint64_t nchunks = n / 1024;
    if (block == 128) hipLaunchKernelGGL(cast_v8<128>, dim3((unsigned)nchunks), dim3(128), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    else hipLaunchKernelGGL(cast_v8<256>, dim3((unsigned)nchunks), dim3(256), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    return (int)hipGetLastError();
  }
  if (variant == 7) {
    const uint64_t ntiles = n / 4096;
    hipLaunchKernelGGL(cast_v7, dim3((unsigned)ntiles), dim3(256), 0, s, (const u32x4*)in, (f32x4*)out, ntiles);
    return (int)hipGetLastError();
  }
  if (variant == 5) {
    const uint64_t nchunks = n / 1024;
    hipLaunchKernelGGL(cast_v5, dim3((unsigned)nchunks), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    return (int)hipGetLastError();
  }
  if (variant == 4) {
    const uint64_t nchunks = n / 1024;
    hipLaunchKernelGGL(cast_v4, dim3((unsigned)nchunks), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    return (int)hipGetLastError();
  }
  if (variant == 2 || variant == 3) {
    const uint64_t nchunks = n / 1024;
    const unsigned grid = block > 0 ? (unsigned)block : (unsigned)(nchunks / (uint64_t)(u > 0 ? u : 1));  // block = grid cap here
    if (variant == 3) hipLaunchKernelGGL(cast_v3, dim3(grid), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    else if (u == 1) hipLaunchKernelGGL(cast_v2<1>, dim3(grid), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    else if (u == 2) hipLaunchKernelGGL(cast_v2<2>, dim3(grid), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    else if (u == 4) hipLaunchKernelGGL(cast_v2<4>, dim3(grid), dim3(64), 0, s, (const u32x4*)in, (f32x4*)out, nchunks);
    else return 1;
    return (int)hipGetLastError();
  }
  if (variant == 0) {
    if (block == 64 && u == 1) GO0(64, 1) else if (block == 64 && u == 4) GO0(64, 4) else if (block == 256 && u == 1) GO0(256, 1)
    else if (block == 256 && u == 4) GO0(256, 4) else if (block == 256 && u == 2) GO0(256, 2) else return 1;
  } else {
    if (block == 64 && u == 1) GO1(64, 1) else if (block == 64 && u == 2) GO1(64, 2) else if (block == 256 && u == 1) GO1(256, 1)
    else if (block == 256 && u == 2) GO1(256, 2) else if (block == 128 && u == 1) GO1(128, 1) else return 1;
  }
  return (int)hipGetLastError();
}
